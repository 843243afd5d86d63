"""One-off scale check of the marking kernels beyond 2^31 elements (64-bit indexing, uint32 indices > 2^31):
hash[i] = (i mod M) * odd  ->  exactly the elements i >= M are duplicates.  python tools/rmdup_scale_check.py [n] [M]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soapnuke_amd import abi  # noqa: E402
from soapnuke_amd.filter import FilterContext  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_300_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1_500_000_000
ctx = FilterContext(abi.default_params(paired=True, max_read_len=150, rmdup=1), device=0)
h = torch.empty(n, dtype=torch.int64, device="cuda")
step = 1 << 28
for a in range(0, n, step):
    b = min(n, a + step)
    i = torch.arange(a, b, dtype=torch.int64, device="cuda")
    h[a:b] = (i % M) * 0x9E3779B97F4A7C15 % (1 << 63)          # (python int wraps are avoided: values stay below 2^63)
    del i
torch.cuda.synchronize()
t0 = time.time()
dup = ctx.mark_dups(h)
torch.cuda.synchronize()
dt = time.time() - t0
bad = 0
for a in range(0, n, step):
    b = min(n, a + step)
    want = (torch.arange(a, b, dtype=torch.int64, device="cuda") >= M).to(torch.uint8)
    bad += int((dup[a:b] != want).sum().item())
print(f"n {n} M {M}: mark {dt:.2f} s ({n / dt / 1e6:.0f} M elements/s), wrong flags {bad}, dups {int(dup.sum(dtype=torch.int64).item())} (expected {n - M})")
sys.exit(1 if bad else 0)
