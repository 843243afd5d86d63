"""One-off scale check of the filter kernel in ONE launch far beyond the bench size (64-bit offsets): R replicas of
1M unique pairs -> every counter must be R x the single-replica counter, every replica's records identical.
python tools/filter_scale_check.py [replicas]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import PE_CASES  # noqa: E402
from soapnuke_amd import abi, synth  # noqa: E402
from soapnuke_amd.filter import FilterContext  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
u = 1_000_000
d = synth.make_batch(u, 150, paired=True, seed=7)
p = abi.default_params(paired=True, max_read_len=150, **PE_CASES["C3_full"])
ctx = FilterContext(p, device=0)
dev = ctx.upload(d)
rec1 = ctx.alloc_records(u)
ctx.filter_batch(ctx.make_batch(dev), rec1)
s1, m1, e1 = ctx.fetch()
ctx.clear()
big = {"n": u * R, "L": 150, "pitch": dev["pitch"], "seq": [x.repeat(R, 1) for x in dev["seq"]], "qual": [x.repeat(R, 1) for x in dev["qual"]], "len": [None, None]}
rec = ctx.alloc_records(u * R)
ctx.filter_batch(ctx.make_batch(big), rec)
s, mx, e = ctx.fetch()
ok = e[0] == 0 and np.array_equal(s, s1 * np.uint64(R))
for m in range(2):
    for k in (0, R // 2, R - 1):
        ok = ok and bool(torch.equal(rec[m][k * u:(k + 1) * u], rec1[m]))
print(f"{u * R} pairs in one launch ({4 * u * R * dev['pitch'] / 2**30:.0f} GiB of planes): {'OK' if ok else 'MISMATCH'}")
sys.exit(0 if ok else 1)
