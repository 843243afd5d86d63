"""tools/isa_cov_run.py <coverage dir> <name>=<spec json>[@ENV=V,ENV=V] ... : captures (tests/isa_interp_capture.py on the emulated library)
and their replays from the kept gfx950 assembly, in parallel, leaving coverage records in <coverage dir> -- the exploration tool behind
the capture lists of tests/test_simt_isa_coverage.py (then: tools/isa_coverage.py <coverage dir>)."""
import concurrent.futures
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT, HERE]


def one(job):
    cov_dir, name, spec, env, kernels = job
    os.environ["SNK_ISA_COV_DIR"] = cov_dir
    import test_simt_isa_interp as TI
    import gfx950_interp as G
    with tempfile.TemporaryDirectory(prefix="isacov_") as tmp:
        launches = TI.capture(tmp, spec, env, kernels=kernels)
        out = []
        for k in launches:
            info, diffs = G.replay(tmp, k, TI.BUILD, verbose=False, garbage=1)
            out.append((info["symbol"][:70], info["instructions"], len(diffs)))
    return name, out


def run(cov_dir, cases, workers=None):
    jobs = [(cov_dir, n, s, e, k) for n, (s, e, k) in cases.items()]
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers or min(8, os.cpu_count() or 1)) as pool:
        return list(pool.map(one, jobs))


if __name__ == "__main__":
    cov = sys.argv[1]
    cases = {}
    for a in sys.argv[2:]:
        name, rest = a.split("=", 1)
        spec, _, env = rest.partition("@")
        cases[name] = (json.loads(spec), dict(kv.split("=") for kv in env.split(",") if kv), ("snk_tiled",))
    for name, out in run(cov, cases):
        print(name, out)
