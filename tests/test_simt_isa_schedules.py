"""The kept gfx950 assembly under OTHER wave schedules (tools/gfx950_interp.py --schedule).  No GPU.

tests/test_simt_isa_interp*.py replay captured launches with the wavefronts of a workgroup taking turns in the emulated twin's own
order: wave 0 up to its barrier, then wave 1, ...  The hardware promises no order between two barriers, and that is where the one bug
of this repository that only a hardware stress run ever found lived (round 3's dormant race on a row register, profiles/r04_ab.txt
item 4): data handed from one wave to another -- through LDS or through memory -- without a barrier in between survives every
schedule in which the producer happens to run first.  Here the same launches run with the waves reversed and under a pre-emptive
random scheduler (a random runnable wave runs 1..Q instructions, then the next draw; workgroups in a shuffled order): a kernel
that is correct whatever the order leaves the same memory, byte for byte, as the emulated twin did.

The launches are chosen so that the barriers MATTER: one workgroup takes several tiles per wave and flushes its LDS histograms
after every trip (SNK_TEST_MAX_WGS=1, SNK_TEST_FLUSH_EVERY=1), so the next trip's adds race with the flush's read-and-clear if
the barrier behind the flush is missing.  The negative controls take barriers out of the assembly and show which schedule sees it --
the ones behind the flush are invisible to the forward order, to the reversed one and to an even-handed random scheduler, and
caught once the waves have paces of their own ("skew": one wave is still in the flush loop while the others add again).

What this still is not: hardware (no timing, no caches, no concurrency between workgroups -- they run one after the other, in a
shuffled order)."""
import concurrent.futures
import json
import os
import re

import pytest

import test_simt_isa_interp as I
from test_simt_isa_interp import G, ASM, needs_asm

# name -> (capture spec, environment of the capture, part of the instance's mangled name that must have run)
CASES = {
    "pe150_c2_two_trips": (dict(case="C2_adatrim_lowq", n=1100, L=150), {"SNK_TEST_MAX_WGS": "1"}, "ILi5ELb0ELb1ELi16ENS_9TileShapeILi160"),
    "pe150_c3_flush_every_trip": (dict(case="C3_full", n=2400, L=150, var_len=True), {"SNK_TEST_MAX_WGS": "1", "SNK_TEST_FLUSH_EVERY": "1"},
                                  "ILi5ELb1ELb1ELi16ENS_9TileShapeILi160"),
    "pe250_c3_flush_every_trip": (dict(case="C3_full", n=1500, L=250, var_len=True), {"SNK_TEST_MAX_WGS": "1", "SNK_TEST_FLUSH_EVERY": "1"},
                                  "ILi8ELb1ELb1ELi16ENS_9TileShapeILi256"),
    "pe150_register_path_two_workgroups": (dict(case="C3_full", n=1500, L=150, pitch=152, var_len=True), {}, "ILi5ELb1ELb0"),
    "se100_c3_two_workgroups": (dict(case="C3_full", n=1500, L=100, paired=False, var_len=True), {}, "ILi4ELb1ELb1"),
}
SCHEDULES = ["reverse", "random:1", "skew:2", "skew:5:15"]
# an ordinary run (tests/conftest.py: SNK_SIMT_FULL=1 takes everything)
CORE = ["test_a_missing_barrier_behind_the_flush"]


def replay_one(args):
    d, k, asm, schedule = args
    try:
        info, diffs = G.replay(d, k, asm, verbose=False, schedule=schedule, garbage=3)      # (and the registers start as noise)
        return k, schedule, info["symbol"], info["instructions"], diffs, None
    except G.Hazard as e:
        return k, schedule, "", 0, [], "Hazard: %s" % e


def replay_all(d, launches, asm, schedules):
    with concurrent.futures.ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        return list(pool.map(replay_one, [(d, k, asm, s) for k in launches for s in schedules]))


@needs_asm
@pytest.mark.parametrize("name", list(CASES))
def test_other_wave_schedules_leave_the_same_memory(name, tmp_path):
    spec, env, instance = CASES[name]
    launches = I.capture(tmp_path, spec, env)
    assert len(launches) >= 2                     # the tiled kernel and the reduce kernel behind it
    results = replay_all(str(tmp_path), launches, ASM, SCHEDULES)
    bad = [(k, s, sym, err or diffs) for k, s, sym, n, diffs, err in results if err or diffs]
    assert not bad, bad
    assert any(instance in sym for _, _, sym, _, _, _ in results), [r[2] for r in results]


@pytest.fixture
def flush_capture(tmp_path):
    spec, env, _ = CASES["pe150_c3_flush_every_trip"]
    I.capture(tmp_path, dict(spec, n=1300), env)
    meta = json.load(open(os.path.join(str(tmp_path), "L0.json")))
    return str(tmp_path), G.symbol_at(meta["lib"], meta["offset"])


SKEWS = ["skew:1", "skew:2", "skew:3", "skew:4"]


@needs_asm
def test_a_missing_barrier_behind_the_flush(tmp_path, flush_capture):
    """__syncthreads_or behind the flush (three s_barrier in the assembly) is what keeps the next trip's histogram adds away from a
    wave that is still reading and clearing the workgroup's words.  Without them no count is lost as long as a wave runs from
    barrier to barrier undisturbed -- forward, reversed: the same memory -- and the skewed pre-emptive scheduler loses counts;
    WITH them the same schedules leave the emulated twin's memory."""
    d, sym = flush_capture

    def edit(body):
        pos = [m.start() for m in re.finditer(r"\ts_barrier\n", body)]
        assert len(pos) == 6, len(pos)            # zeroing | before the flush | __syncthreads_or: 3 | before the trimming counters' drain
        for p in reversed(pos[2:5]):
            body = body[:p] + "\ts_nop 0\n" + body[p + len("\ts_barrier\n"):]
        return body

    broken = I.mutated(tmp_path, sym, edit)
    full = os.environ.get("SNK_SIMT_FULL") == "1"
    jobs = [(d, 0, ASM, s) for s in (SKEWS if full else SKEWS[:2])] + [(d, 0, broken, s) for s in SKEWS]
    if full:
        jobs += [(d, 0, broken, None), (d, 0, broken, "reverse")]
    with concurrent.futures.ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:      # (one batch: ~ 20 s each)
        results = list(zip(jobs, pool.map(replay_one, jobs)))
    good = [r for j, r in results if j[2] == ASM]
    bad = [r for j, r in results if j[2] == broken and j[3] in SKEWS]
    calm = [r for j, r in results if j[2] == broken and j[3] not in SKEWS]
    assert not [r for r in good if r[4] or r[5]], good
    assert any(r[4] for r in bad), bad
    # (not asserted, observed in a full run: the undisturbed orders do not see it -- [False, False])
    print("without the barriers: forward / reversed differ:", [bool(r[4]) for r in calm], "skewed:", [bool(r[4]) for r in bad])


@needs_asm
def test_skewed_paces_leave_the_same_memory(tmp_path):
    """the flush-every-trip launch of 2400 ragged pairs (three trips of sixteen waves) and its reduce kernel under four skewed schedules"""
    spec, env, instance = CASES["pe150_c3_flush_every_trip"]
    launches = I.capture(tmp_path, spec, env)
    results = replay_all(str(tmp_path), launches, ASM, SKEWS)
    bad = [(k, s, sym, err or diffs) for k, s, sym, n, diffs, err in results if err or diffs]
    assert not bad, bad
    assert any(instance in r[2] for r in results)
