"""Oracle (and, with -m gpu, the HIP path) against the committed golden vectors that
tests/golden/make_golden.py produced from the compiled reference.  Does not need
/root/reference or oracle/_ref at run time."""
import hashlib
import json
import os

import numpy as np
import pytest

import snk_testlib as T
from cases import PE_CASES, se_kwargs
from soapnuke_amd import abi, synth

G = os.path.join(T.ROOT, "tests", "golden")
INDEX = json.load(open(os.path.join(G, "index.json")))["vectors"]
_cache = {}


def _inputs(v):
    key = (v["shape"],)
    if key not in _cache:
        d = synth.make_batch(v["n"], v["L"], paired=v["paired"], var_len=v["var_len"], seed=v["seed"])
        h = hashlib.sha256(b"".join(np.ascontiguousarray(x).tobytes() for x in d["seq"] + d["qual"])).hexdigest()
        if h != v["input_sha256"]:
            if v["shape"] != "pe150":
                pytest.skip("generator output changed; regenerate tests/golden (input_pe150.npz still checked)")
            z = np.load(os.path.join(G, "input_pe150.npz"))
            d = dict(n=v["n"], L=150, pitch=z["seq1"].shape[1], paired=True, seq=[z["seq1"], z["seq2"]],
                     qual=[z["qual1"], z["qual2"]], len=[None, None])
        _cache[key] = d
    return _cache[key]


def _expected(v, p):
    z = np.load(os.path.join(G, v["file"]))
    s, _ = T.new_stats(p)
    s[z["sum_idx"]] = z["sum_val"]
    dt = abi.record_dtype()
    return dict(rec=[z["rec1"].copy().view(dt).reshape(-1), z["rec2"].copy().view(dt).reshape(-1)], sum=s, max=z["max"])


def _params(v):
    kw = PE_CASES[v["case"]] if v["paired"] else se_kwargs(PE_CASES[v["case"]])
    return abi.default_params(paired=v["paired"], max_read_len=v["L"], **kw)


def _check(got, want, p, paired):
    for m in range(2 if paired else 1):
        assert np.array_equal(got["rec"][m], want["rec"][m]), f"records differ, mate {m}"
    assert np.array_equal(got["sum"], want["sum"]), T.describe_stats_diff(p, got["sum"], want["sum"])
    assert np.array_equal(got["max"], want["max"])


def test_stored_input_matches_generator():
    v = next(x for x in INDEX if x["shape"] == "pe150")
    z = np.load(os.path.join(G, "input_pe150.npz"))
    h = hashlib.sha256(b"".join(z[k].tobytes() for k in ("seq1", "seq2", "qual1", "qual2"))).hexdigest()
    assert h == v["input_sha256"]


@pytest.mark.parametrize("v", INDEX, ids=[x["file"][:-4] for x in INDEX])
def test_oracle_matches_golden(v):
    p = _params(v)
    _check(T.run_oracle(p, _inputs(v)), _expected(v, p), p, v["paired"])


@pytest.mark.gpu
@pytest.mark.parametrize("v", INDEX, ids=[x["file"][:-4] for x in INDEX])
def test_hip_matches_golden(v):
    from test_gpu_parity import run_hip_device
    p = _params(v)
    got = run_hip_device(p, _inputs(v), 0)
    assert got["err"][0] == 0
    _check(got, _expected(v, p), p, v["paired"])
