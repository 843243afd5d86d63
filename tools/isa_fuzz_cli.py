"""tools/isa_fuzz_cli.py <first seed> <count> : random runs of the EMULATED CLI with every kernel launch captured and replayed from the kept
gfx950 assembly (the harness of tests/test_simt_isa_interp_cli.py: the first launches of every kernel, memory byte for byte, no result
used ahead of its s_waitcnt; first-come-first-placed tables compared by content).  Read length, pairing, .gz in / out, the inflate on the
device (cooperative or lane-0), duplicate marking in one or two passes, contaminant lists, batch sizes, the command line's thresholds: drawn
per seed.  What the run WRITES is compared with the reference binary by tests/test_simt_cli_fuzz.py; this compares the instructions with
their twins for the kernels tools/isa_fuzz.py does not reach (FASTQ index / scatter / format and their scans, inflate, deflate, hashing)."""
import concurrent.futures
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT, HERE]


def scenario(seed):
    import numpy as np
    import test_simt_isa_interp_cli as TC
    from soapnuke_amd import synth
    rng = np.random.default_rng(88000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]       # noqa: E731
    paired = bool(rng.random() < 0.65)
    L = pick([100, 150, 150, 250, 400])
    gz_in, gz_out = bool(rng.random() < 0.6), bool(rng.random() < 0.5)
    cfg, env = [], {"SNK_BATCH_PAIRS": pick(["64", "96", "128"])}
    cli = ["-f", synth.ADAPTER1] + (["-r", synth.ADAPTER2] if paired else []) + ["-l", str(pick([5, 10, 20])), "-q", str(pick([0.1, 0.2, 0.5]))]
    if rng.random() < 0.6:
        cli.append("-J")
    if rng.random() < 0.5:
        cli += ["-n", "0.01", "-m", "20", "-g", "10", "-X", "50", "-p", "0.8"]
    if rng.random() < 0.4:
        cfg.append("rmdup")
        if rng.random() < 0.4:
            env["SNK_RMDUP_TWO_PASS"] = "1"
    if rng.random() < 0.25 and L <= 250:
        cfg += TC.CONTAM_CFG + (["contam2=" + TC.CT2] if paired else [])
    if gz_in and rng.random() < 0.7:
        env.update(SNK_DEVICE_INFLATE="1", SNK_DGZ_WINDOW_MB="1", SNK_DGZ_CHUNK_KB=pick(["8", "16"]))
        if rng.random() < 0.4:
            env["SNK_DGZ_COOP"] = "0"
    if rng.random() < 0.15:
        env["SNK_PROVEN_ONLY"] = "1"
    return dict(n=int(rng.integers(100, 320)), L=L, paired=paired, gz_in=gz_in, gz_out=gz_out, cfg=cfg, cli=cli, env=env, expect=[],
                gz_level=pick([1, 1, 4, 6, 9]), gz_members=pick([1, 1, 2, 3]), gz_stored=bool(rng.random() < 0.2))


def main():
    import test_simt_isa_interp_cli as TC
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad_total = 0
    for seed in range(first, first + count):
        sc = scenario(seed)
        with tempfile.TemporaryDirectory(prefix="isafuzzcli_") as tmp:
            try:
                dump, launches = TC.capture_run(tmp, sc)
                with concurrent.futures.ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
                    results = list(pool.map(TC.replay_one, [(dump, k, None) for k in launches]))
                bad = [(k, sym[:60], err or diffs) for k, sym, n, diffs, err in results if err or diffs]
                kernels = sorted({sym.split("N_1")[-1][:28] for _, sym, n, _, _ in results if n > 0})
            except Exception as ex:          # noqa: BLE001
                bad, kernels = [("capture", type(ex).__name__, str(ex)[-500:])], []
        desc = "L=%d %s %s->%s %s" % (sc["L"], "PE" if sc["paired"] else "SE", "gz" if sc["gz_in"] else "plain", "gz" if sc["gz_out"] else "plain",
                                    " ".join(c.split("=")[0] for c in sc["cfg"][:2]) + " " + " ".join("%s=%s" % kv for kv in sorted(sc["env"].items()) if kv[0] != "SNK_BATCH_PAIRS"))
        print("seed %d %s: %s" % (seed, desc, ("%d kernels identical" % len(kernels)) if not bad else "DIFFERS %r" % (bad[:3],)), flush=True)
        if bad:
            print("   scenario:", json.dumps({k: v for k, v in sc.items()}), flush=True)
            bad_total += 1
    print("%d of %d runs differ" % (bad_total, count))
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
