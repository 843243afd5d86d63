"""Times the rmdup pre-pass kernels on resident data: python tools/bench_rmdup.py [pairs] [L]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soapnuke_amd import abi, synth  # noqa: E402
from soapnuke_amd.filter import FilterContext  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
uniq = min(n, 1_000_000)
d = synth.make_batch(uniq, L, paired=True)
ctx = FilterContext(abi.default_params(paired=True, max_read_len=L, rmdup=1), device=0)
dev = ctx.upload(d)
reps = n // uniq
dev["seq"] = [x.repeat(reps, 1) for x in dev["seq"]]
dev["qual"] = [x.repeat(reps, 1) for x in dev["qual"]]
dev["n"] = uniq * reps
b = ctx.make_batch(dev)
h = ctx.hash_batch(b)
torch.cuda.synchronize()


def timed(f, k=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        r = f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k, r


ms_h, _ = timed(lambda: ctx.hash_batch(b, h))
# realistic duplicate structure for marking: 5 % duplicates, otherwise distinct (replicas would all collide)
hh = torch.randint(-2**62, 2**62, (dev["n"],), dtype=torch.int64, device="cuda")
k = dev["n"] // 20
hh[torch.randint(0, dev["n"], (k,), device="cuda")] = hh[torch.randint(0, dev["n"], (k,), device="cuda")]
ms_m, dup = timed(lambda: ctx.mark_dups(hh))
nn = dev["n"]
print(f"pairs {nn} L {L}: hash {ms_h:.3f} ms ({nn / ms_h / 1e3:.1f} Mpairs/s, {2 * L * nn / ms_h / 1e6:.0f} GB/s of sequence bytes); "
      f"mark {ms_m:.3f} ms ({nn / ms_m / 1e3:.1f} Mpairs/s), dups {int(dup.sum())}")
