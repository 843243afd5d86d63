"""tools/isa_fuzz.py <first seed> <count> [workers] : random parameter contexts through the instruction tier.

For every seed: a read length / pitch / pairing / raggedness and a random parameter set the reference defines (tests/cases.py::random_context:
adapter lists, mismatch budgets, ratios, hard and low-quality trims, length limits, poly-G / poly-X ...), one batch through the EMULATED
library with the launches captured (tests/isa_interp_capture.py compares the emulated result with the oracle), every captured launch
replayed from the kept gfx950 assembly (tools/gfx950_interp.py, registers start as noise) and compared byte for byte with the twin.
Prints one line per seed; exit code 1 when any replay differs, trips a hazard or meets an instruction the interpreter refuses.

Round 6: the hand-made captures found the interpreter ignoring the integer `clamp`; this is the sweep for whatever else only an odd
parameter set reaches -- in the interpreter or in the compiled code."""
import concurrent.futures
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT, HERE]

SHAPES = [(150, None), (250, None), (100, None), (50, None), (180, None), (200, None), (140, None), (150, 152), (100, 104), (64, None), (33, None), (256, None),
          (300, None), (600, None), (1000, None), (150, None), (250, None)]      # (beyond 256 positions: the long-read kernels)


def context(seed):
    import numpy as np
    from cases import random_context
    rng = np.random.default_rng(77000 + seed)
    L, pitch = SHAPES[int(rng.integers(0, len(SHAPES)))]
    paired = bool(rng.random() < 0.7)
    var_len = bool(rng.random() < 0.6)
    min_len = L if not var_len else max(L // 2, 20)
    if L < 40:
        var_len, min_len = False, L
    kw = random_context(rng, L, paired, min_len)
    if not paired:
        kw = {k: v for k, v in kw.items() if k != "adapters2"}
    kw = json.loads(json.dumps(kw))                      # (numpy scalars -> plain numbers)
    spec = dict(case="defaults", n=int(rng.choice([64, 130, 200, 257])), L=L, paired=paired, var_len=var_len, seed=int(rng.integers(1, 10000)), kw=kw,
                dimer_frac=float(rng.choice([0.0, 0.1, 0.3])), kernel=2)
    if pitch:
        spec["pitch"] = pitch
    if rng.random() < 0.3:
        spec["lower"] = float(rng.choice([0.05, 0.2]))
    if rng.random() < 0.2:
        spec["dup"] = 1
        spec["kw"]["rmdup"] = 1
    if L > 256:
        spec["kernel"], spec["n"] = 0, min(spec["n"], 96)
    if os.environ.get("SNK_ISA_FUZZ_CONTAM") == "1":    # every context with RANDOM contaminant lists (lengths, ratios, budgets, global contaminants)
        B = "ACGT"
        word = lambda a, z: "".join(B[int(x)] for x in rng.integers(0, 4, int(rng.integers(a, z))))      # noqa: E731
        hi = min(64, max(12, (L if not var_len else L // 2) - 4))
        ck = {}
        if rng.random() < 0.8:
            k = int(rng.integers(1, 3))
            ck["contam1"] = ",".join(word(8, hi) for _ in range(k))
            if paired:
                ck["contam2"] = ",".join(word(8, hi) for _ in range(k))
            ck["ct_match_r"] = ",".join(str(rng.choice([0.2, 0.4, 0.5, 0.7, 0.9, 1.0])) for _ in range(k)) if k > 1 else str(rng.choice([0.2, 0.5, 0.8, 1.0]))
        if rng.random() < 0.6 or not ck:
            k = int(rng.integers(1, 3))
            ck["global_contams"] = ",".join(word(12, hi) for _ in range(k))
            ck["g_mrs"] = ",".join(str(rng.choice([0.3, 0.5, 0.7, 1.0])) for _ in range(k))
            ck["g_mms"] = ",".join(str(int(rng.integers(0, 5))) for _ in range(k))
        if rng.random() < 0.5:
            ck["contam_trim"] = 1
        ck["ada_mis"] = [int(rng.integers(0, 6)), int(rng.integers(0, 6))]
        ck["ada_edge"] = [int(rng.integers(1, 8)), int(rng.integers(1, 8))]
        for k in ("ada_mis", "ada_edge"):
            spec["kw"].pop(k, None)
        spec["contam"] = json.loads(json.dumps(ck))
        spec["kernel"] = 0 if L > 256 else 2
    elif rng.random() < 0.25:                            # contaminant lists on top (their kernels run in front of the tiled / long-read decide kernel)
        from cases import CONTAM_CASES
        spec["contam"] = str(rng.choice(sorted(CONTAM_CASES)))
        spec["kernel"] = 0 if L > 256 else 2
    return spec


def one(seed):
    import gfx950_interp as G
    import test_simt_isa_interp as TI
    spec = context(seed)
    out = []
    # SNK_ISA_FUZZ_SCHEDULE=skew|random|reverse: the waves of a workgroup take turns under that adversarial scheduler (tools/gfx950_interp.py
    # run_launch) -- and the batch is made one where the barriers matter: ONE workgroup, several trips per wave, the LDS histograms flushed
    # after every trip (the test hooks of csrc/snk_tiled.hip)
    sched, env = os.environ.get("SNK_ISA_FUZZ_SCHEDULE"), {}
    if sched:
        env = {"SNK_TEST_MAX_WGS": "1", "SNK_TEST_FLUSH_EVERY": "1"}
        if spec["L"] <= 256:
            spec["n"] = 1500 if spec["L"] <= 150 else 1100
        sched = sched if sched == "reverse" else "%s:%d" % (sched, seed)
    try:
        with tempfile.TemporaryDirectory(prefix="isafuzz_") as tmp:
            launches = TI.capture(tmp, spec, env, kernels=("snk_tiled", "snk_long", "snk_contam"))
            for k in launches:
                info, diffs = G.replay(tmp, k, TI.BUILD, verbose=False, garbage=seed, schedule=sched)
                bad = bool(diffs) or bool(info["scalar_loads_of_words_written_in_this_launch"])
                out.append((info["symbol"][22:62], info["instructions"], "DIFFERS %r" % (diffs[:2],) if bad else "ok"))
    except Exception as ex:              # noqa: BLE001 -- a capture the emulated tier rejects, a hazard, an unknown instruction: all findings
        out.append(("-", 0, "EXCEPTION %s: %s" % (type(ex).__name__, str(ex)[-400:])))
    return seed, spec, out


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else min(8, os.cpu_count() or 1)
    import test_simt_isa_interp as TI
    TI.simt_lib_path()
    bad = 0
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers) as pool:
        for seed, spec, out in pool.map(one, range(first, first + count)):
            ok = all(o[2] == "ok" for o in out) and out
            bad += 0 if ok else 1
            print("seed %d L=%d%s %s %s n=%d: %s" % (seed, spec["L"], "/%d" % spec["pitch"] if "pitch" in spec else "", "PE" if spec["paired"] else "SE",
                                                    "ragged" if spec["var_len"] else "fixed", spec["n"],
                                                    "identical (%s)" % ", ".join(o[0].split("ELi16")[0] for o in out) if ok else out), flush=True)
            if not ok:
                print("   spec:", json.dumps(spec), flush=True)
    print("%d of %d contexts differ" % (bad, count))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
