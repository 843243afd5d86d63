"""Every kernel a run of the CLI launches, replayed from the gfx950 assembly the build keeps (tools/gfx950_interp.py).

tests/test_simt_isa_interp.py does this for the tiled kernel's instances on launches of the C ABI.  Here the EMULATED CLI
(tests/simt: soapnuke_amd/host against the emulated library) runs whole scenarios with SIMT_DUMP_DIR set -- FASTQ ingest, inflate,
the filter kernels, duplicate marking, formatting, gzip --, the first launch(es) of every kernel are captured (kernarg segment and
all device memory before / after; while snapshots are taken the launches and copies of all host threads take turns), and each one
is run again from the kernel's gfx950 instructions on the captured state: the memory it leaves must be the emulated twin's, byte for
byte, and no result may be used before the s_waitcnt that covers it.  What the emulated CLI writes is compared with the reference
binary by tests/test_simt_cli.py; this file compares the instructions with the emulated twins.

First run of this file (round 5): 42 kernels, all identical -- the device inflate (search / lane-0 decode / cooperative decode /
chain / resolve), the FASTQ index / scatter / format kernels and their scans, the deflate kernels with their run-time constant
tables (hipMemcpyToSymbol'd device variables reached through the global offset table), hash / insert / look-up of one- and
two-pass duplicate marking, the contaminant and long-read kernels, the generic kernel.  No GPU."""
import concurrent.futures
import gzip
import json
import os
import subprocess
import sys

import pytest

import simt_lib as S
import snk_testlib as T
from soapnuke_amd import synth

sys.path.insert(0, os.path.join(T.ROOT, "tools"))
import gfx950_interp as G          # noqa: E402

BUILD = os.path.join(T.ROOT, "soapnuke_amd", "csrc", "build")
ADAPTERS = ["-f", synth.ADAPTER1, "-r", synth.ADAPTER2]
from cases import CT1, CT2, GC1, plant_contams          # noqa: E402

CONTAM_KW = dict(contam1=CT1 + ",GGGGGGGGGGGGGGGGGGGGGGGG", contam2=CT2, global_contams=GC1)
CONTAM_CFG = ["contam1=" + CONTAM_KW["contam1"], "ctMatchR=0.6,0.7", "global_contams=" + GC1, "glob_cotm_mR=0.4", "glob_cotm_mM=1"]

# name -> dict(n, L, paired, gz_in, gz_out, cfg lines, extra command line, environment, kernels that must have been replayed)
SCENARIOS = {
    # the README shape: .gz in, .gz out, duplicates marked in one pass, the device decodes the input (cooperative decode)
    "pe150_gz_rmdup_device_inflate": dict(
        n=128, L=150, paired=True, gz_in=True, gz_out=True, cfg=["rmdup"], cli=ADAPTERS + ["-J", "-l", "10", "-q", "0.1"],
        env={"SNK_DEVICE_INFLATE": "1", "SNK_DGZ_WINDOW_MB": "1", "SNK_DGZ_CHUNK_KB": "8", "SNK_BATCH_PAIRS": "64"},
        expect=["inf_search_kernel", "inf_decode_coop_kernel", "inf_chain_kernel", "inf_resolve_kernel", "fq_count_kernel", "fq_index_kernel",
                "scan_sums_kernel", "scan_top_kernel", "scan_apply_kernel", "fq_scatter_kernel", "snk_hash_lds_kernel", "snk_stream_insert_kernel",
                "snk_stream_lookup_kernel", "snk_tiled_kernel", "snk_tiled_reduce_kernel", "fq_outlen_kernel", "fq_format_kernel", "dfl_hist_kernel",
                "dfl_build_kernel", "dfl_bits_kernel", "dfl_member_kernel", "dfl_emit_kernel", "snk_finalize_kernel"]),
    # single end, the lane-0 decoder, two passes of duplicate marking, a contaminant list on 150-position reads
    "se150_gz_two_pass_rmdup_contam": dict(
        n=300, L=150, paired=False, gz_in=True, gz_out=False, cfg=["rmdup"] + CONTAM_CFG, cli=["-f", synth.ADAPTER1, "-l", "10", "-q", "0.2"],
        env={"SNK_DEVICE_INFLATE": "1", "SNK_DGZ_COOP": "0", "SNK_DGZ_WINDOW_MB": "1", "SNK_DGZ_CHUNK_KB": "8", "SNK_BATCH_PAIRS": "128", "SNK_RMDUP_TWO_PASS": "1"},
        expect=["inf_decode_kernel", "snk_mark_insert_kernel", "snk_mark_lookup_kernel", "snk_contam_kernel", "snk_tiled_kernel"]),
    # reads of 400 positions: the long-read path (prep / decide / histograms) and its contaminant kernel
    "pe400_long_reads_contam": dict(
        n=120, L=400, paired=True, gz_in=False, gz_out=False, cfg=CONTAM_CFG + ["contam2=" + CT2], cli=ADAPTERS + ["-J", "-l", "10", "-q", "0.2"],
        env={"SNK_BATCH_PAIRS": "64"},
        expect=["snk_long_prep_kernel", "snk_long_decide_kernel", "snk_long_hist_kernel", "snk_long_contam_kernel"]),
    # outside the envelope SNK_PROVEN_ONLY=1 keeps for the tiled kernel (a five-character adapter): the generic kernel decides
    "pe150_generic_kernel": dict(
        n=200, L=150, paired=True, gz_in=False, gz_out=False, cfg=[], cli=["-f", "AAGTC", "-r", "AAGTC", "-l", "10", "-q", "0.2"],
        env={"SNK_PROVEN_ONLY": "1", "SNK_BATCH_PAIRS": "128"},
        expect=["snk_generic_kernel"]),
    # 250 positions: the eight-plane-word instances of the tiled and the contaminant kernel (BASELINE configs[4]'s shape)
    "pe250_contam_rmdup": dict(
        n=160, L=250, paired=True, gz_in=False, gz_out=True, cfg=["rmdup"] + CONTAM_CFG + ["contam2=" + CT2],
        cli=ADAPTERS + ["-J", "-l", "10", "-q", "0.2", "-n", "0.01", "-m", "20", "-g", "10", "-X", "50", "-p", "0.8"],
        env={"SNK_BATCH_PAIRS": "96"},
        expect=["snk_tiled_kernel", "snk_contam_kernel", "snk_hash_lds_kernel"]),
    # single end, duplicates marked in one pass: the shift kernel that moves a batch's flags by one read, the single-end hash kernel
    "se150_one_pass_rmdup": dict(
        n=300, L=150, paired=False, gz_in=False, gz_out=False, cfg=["rmdup"], cli=["-f", synth.ADAPTER1, "-l", "10", "-q", "0.2"],
        env={"SNK_BATCH_PAIRS": "128"},
        expect=["snk_se_shift_kernel", "snk_hash_lds_kernelILb0E", "snk_stream_insert_kernel", "snk_stream_lookup_kernel"]),
    # two shards on two (emulated) devices, the duplicate table sharded by hash % 2: owner count / scatter, the flags' way home
    "pe150_sharded_rmdup_two_shards": dict(
        n=400, L=150, paired=True, gz_in=False, gz_out=False, cfg=["rmdup"], cli=ADAPTERS + ["-J", "-l", "10", "-q", "0.2", "--devices", "0,1"],
        env={"SNK_SHARDED": "1", "SNK_SHARD_WIRE": "host", "SNK_SHARD_MIN_RECORDS": "10", "SIMT_DEVICES": "2", "SIMT_DUMP_BY_PID": "1", "SNK_BATCH_PAIRS": "128"},
        expect=["snk_owner_count_kernel", "snk_owner_scatter_kernel", "snk_flags_home_kernel", "snk_tiled_kernel"]),
    # reads of 1000 positions with duplicate marking: the 64-lane prep instance, rows too long for the LDS-staged hash kernel
    "pe1000_rmdup": dict(
        n=96, L=1000, paired=True, gz_in=False, gz_out=False, cfg=["rmdup"], cli=ADAPTERS + ["-J", "-l", "10", "-q", "0.3"],
        env={"SNK_BATCH_PAIRS": "64"},
        expect=["snk_long_prep_kernelILi64E", "snk_hash_direct_kernelILb1E", "snk_long_decide_kernel", "snk_long_hist_kernel"]),
    "se1000_rmdup": dict(
        n=96, L=1000, paired=False, gz_in=False, gz_out=False, cfg=["rmdup"], cli=["-f", synth.ADAPTER1, "-l", "10", "-q", "0.3"],
        env={"SNK_BATCH_PAIRS": "64"},
        expect=["snk_hash_direct_kernelILb0E"]),
}
# an ordinary run (tests/conftest.py: SNK_SIMT_FULL=1 takes everything)
CORE = ["test_every_kernel_of_a_run_matches_its_emulated_twin[pe150_gz_rmdup_device_inflate]"]
pytestmark = pytest.mark.skipif(not os.path.isdir(BUILD), reason="the build's kept assembly is not there (python __graft_entry__.py)")


def write_inputs(work, sc):
    n, L, paired = sc["n"], sc["L"], sc["paired"]
    d = synth.make_batch(n, L, paired=paired, seed=17, var_len=True, dimer_frac=0.1)
    rows = max(8, n // 10)
    if any(c.startswith("contam1=") for c in sc["cfg"]):       # whole / truncated copies of the contaminants in some reads
        plant_contams(d, {k: v for k, v in CONTAM_KW.items() if paired or k != "contam2"})
    for m in range(2 if paired else 1):                       # duplicates
        for key in ("seq", "qual"):
            d[key][m][n // 2:n // 2 + rows] = d[key][m][0:rows]
        if d["len"][m] is not None:
            d["len"][m][n // 2:n // 2 + rows] = d["len"][m][0:rows]
    files = []
    for m in range(2 if paired else 1):
        f = os.path.join(work, "r%d.fq" % (m + 1))
        synth.write_fastq(f, d["seq"][m], d["qual"][m], L, m + 1, lens=d["len"][m])
        if sc["gz_in"]:
            # (tools/isa_fuzz_cli.py: other compression levels, several members, a stored member in between)
            raw, level, members = open(f, "rb").read(), int(sc.get("gz_level", 6)), int(sc.get("gz_members", 1))
            cuts = [len(raw) * k // members for k in range(members + 1)]
            with open(f + ".gz", "wb") as dst:
                for k in range(members):
                    dst.write(gzip.compress(raw[cuts[k]:cuts[k + 1]], compresslevel=0 if (sc.get("gz_stored") and k == 1) else level))
            f += ".gz"
        files.append(f)
    return files


def capture_run(work, sc, per_kernel=1):
    cli = S.build_module().build_cli()
    files = write_inputs(work, sc)
    dump = os.path.join(work, "dump")
    os.makedirs(dump)
    ext = ".fq.gz" if sc["gz_out"] else ".fq"
    cmd = [cli, "filter", "-1", files[0], "-C", "c1" + ext, "-o", os.path.join(work, "out"), "-T", "1"]
    if sc["paired"]:
        cmd += ["-2", files[1], "-D", "c2" + ext]
    if sc["cfg"]:
        with open(os.path.join(work, "cfg"), "w") as f:
            f.write("\n".join(sc["cfg"]) + "\n")
        cmd += ["-c", os.path.join(work, "cfg")]
    env = dict(os.environ, SIMT_DUMP_DIR=dump, SIMT_DUMP_PER_KERNEL=str(per_kernel), SIMT_CUS="2", **sc["env"])
    r = subprocess.run(cmd + sc["cli"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    return dump, sorted(int(f[1:-5]) for f in os.listdir(dump) if f.endswith(".json"))


# the twin's own order of the waves, and -- in a full run -- a pre-emptive scheduler with a pace per wave (tests/test_simt_isa_schedules.py)
SCHEDULES = [None, "skew:1"] if os.environ.get("SNK_SIMT_FULL") == "1" else [None]


# Open-addressing tables are filled first come, first placed: under another order of waves or workgroups the same keys sit in other slots.
# For these kernels the comparison is the table's CONTENT -- the set of (key, smallest index) pairs -- not its layout; what is read
# from the table (the look-up kernels' flags) is compared byte for byte on those kernels' own launches.
TABLE_FILLERS = ("snk_mark_insert_kernel", "snk_stream_insert_kernel")


def same_table_content(differing):
    """differing: [(base, got, want)] of the two allocations of a table (u64 keys, u32 smallest indices; 2^k slots each)"""
    if len(differing) != 2:
        return False
    (_, ka, kb), (_, ia, ib) = sorted(differing, key=lambda d: -d[1].size)
    cap = 1 << ((ka.size // 8).bit_length() - 1)
    if (ia.size // 4) < cap:
        return False
    import numpy as np
    pairs = []
    for keys, idx in ((ka, ia), (kb, ib)):
        k64 = keys[:cap * 8].view(np.uint64)
        i32 = idx[:cap * 4].view(np.uint32)
        used = k64 != np.uint64(0xFFFFFFFFFFFFFFFF)
        pairs.append(sorted(zip(k64[used].tolist(), i32[used].tolist())))
    return pairs[0] == pairs[1] and len(pairs[0]) > 0 and bytes(ka[cap * 8:]) == bytes(kb[cap * 8:]) and bytes(ia[cap * 4:]) == bytes(ib[cap * 4:])


def same_scatter_content(dump, k, differing):
    """snk_owner_scatter_kernel hands out places behind a per-owner cursor (one atomicAdd per wave and owner): WHICH place a read gets
    depends on the order the waves arrive in.  What must hold whatever the order: every read's slot points at its own (hash, index)
    record, the records are the same set, and every place belongs to the same owner (hash % world) as in the twin's layout."""
    import struct
    import numpy as np
    meta = json.load(open(os.path.join(dump, "L%d.json" % k)))
    ka = bytes.fromhex(meta["kernarg"])
    n, world = struct.unpack_from("<qI", ka, 8)
    p_hash, p_index, p_slot = struct.unpack_from("<QQQ", ka, 40)
    by_base = {b: (got, want) for b, got, want in differing}
    if set(by_base) - {p_hash, p_index, p_slot} or p_hash not in by_base or p_slot not in by_base:
        return False
    gh, wh = (x[:8 * n].view(np.uint64) for x in by_base[p_hash])
    gs, ws = (x[:4 * n].view(np.uint32).astype(np.int64) for x in by_base[p_slot])
    if p_index in by_base:
        gi, wi = (x[:4 * n].view(np.uint32) for x in by_base[p_index])
    else:
        return False
    if sorted(gs.tolist()) != list(range(n)) or sorted(ws.tolist()) != list(range(n)):      # (one shard's launch: the places are 0 .. n-1)
        return False
    return bool((gh[gs] == wh[ws]).all() and (gi[gs] == wi[ws]).all() and ((gh % np.uint64(world)) == (wh % np.uint64(world))).all())


def replay_one(args):
    dump, k, schedule = args
    try:
        info, diffs = G.replay(dump, k, BUILD, verbose=False, schedule=schedule, keep_memory=True,
                               garbage=None if schedule is None else 5)      # (the second pass: other wave order, registers start as noise)
        # (the first pass as well: the emulated twin runs the workgroups of a launch on two OS threads, so the layout IT left is one of
        #  several -- seen once in a full run, with two workgroups' keys meeting in one slot)
        if diffs and any(t in info["symbol"] for t in TABLE_FILLERS) and same_table_content(info.get("differing", [])):
            diffs = []
        if diffs and "snk_owner_scatter_kernel" in info["symbol"] and same_scatter_content(dump, k, info.get("differing", [])):
            diffs = []
        info.pop("differing", None)
        if info["scalar_loads_of_words_written_in_this_launch"]:      # the scalar cache is not coherent with vector stores inside a launch
            diffs = list(diffs) + [("scalar loads of words written in this launch", info["scalar_loads_of_words_written_in_this_launch"])]
        return k, info["symbol"], info["instructions"], diffs, None
    except Exception as e:                                    # (a hazard, an unknown instruction: the test names the kernel)
        meta = json.load(open(os.path.join(dump, "L%d.json" % k)))
        return k, G.symbol_at(meta["lib"], meta["offset"]), 0, [], "%s: %s" % (type(e).__name__, e)


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_every_kernel_of_a_run_matches_its_emulated_twin(name, tmp_path):
    sc = SCENARIOS[name]
    dump, launches = capture_run(str(tmp_path), sc)
    assert launches
    with concurrent.futures.ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(replay_one, [(dump, k, s) for k in launches for s in SCHEDULES]))
    bad = [(k, sym, err or diffs) for k, sym, n, diffs, err in results if err or diffs]
    assert not bad, bad
    seen = [sym for _, sym, n, _, _ in results if n > 0]
    for want in sc["expect"]:
        assert any(want in s for s in seen), (want, seen)
