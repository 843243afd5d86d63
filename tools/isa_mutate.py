"""tools/isa_mutate.py -- a mutation score for the instruction tier (VERDICT r5 "Next round" 3).

The instruction tier (tools/gfx950_interp.py) replays captured launches from the gfx950 assembly the build keeps and compares the memory
a kernel leaves with its emulated twin's.  How much would it notice?  This tool changes ONE instruction of a kernel at a time -- an
opcode swapped for its opposite (add / sub, and / or, shift left / right, min / max, a compare or a branch condition inverted) -- in
instructions the captures EXECUTE, replays the captures that execute it, and counts the mutant as killed when any replay leaves
different memory, trips a hazard, faults or never ends.  Mutants are drawn with a seed; survivors are listed with their text and line.

Library (tests/test_simt_isa_coverage.py drives it): mutants(), run_mutants().  The file is mutated in memory -- one parsed program per
worker process, the instruction object replaced and put back."""
import concurrent.futures
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gfx950_interp as G          # noqa: E402

# opcode -> its mutant.  Only swaps that keep the operand shapes legal for the interpreter's handlers.
SWAPS = {
    "v_add_u32": "v_sub_u32", "v_sub_u32": "v_add_u32", "v_or_b32": "v_and_b32", "v_and_b32": "v_or_b32", "v_xor_b32": "v_or_b32",
    "v_lshlrev_b32": "v_lshrrev_b32", "v_lshrrev_b32": "v_lshlrev_b32", "v_min_u32": "v_max_u32", "v_max_u32": "v_min_u32",
    "v_min_i32": "v_max_i32", "v_max_i32": "v_min_i32",
    "s_add_i32": "s_sub_i32", "s_sub_i32": "s_add_i32", "s_add_u32": "s_sub_u32", "s_sub_u32": "s_add_u32",
    "s_and_b32": "s_or_b32", "s_or_b32": "s_and_b32", "s_and_b64": "s_or_b64", "s_or_b64": "s_and_b64",
    "s_lshl_b32": "s_lshr_b32", "s_lshr_b32": "s_lshl_b32", "s_min_i32": "s_max_i32", "s_max_i32": "s_min_i32",
    "s_cbranch_scc0": "s_cbranch_scc1", "s_cbranch_scc1": "s_cbranch_scc0", "s_cbranch_vccz": "s_cbranch_vccnz", "s_cbranch_vccnz": "s_cbranch_vccz",
    "s_cbranch_execz": "s_cbranch_execnz", "s_cbranch_execnz": "s_cbranch_execz",
}
CMP = {"lt": "ge", "ge": "lt", "gt": "le", "le": "gt", "eq": "ne", "ne": "eq", "lg": "eq"}
CMP_RE = re.compile(r"^(v_cmpx?|s_cmp)_(lt|ge|gt|le|eq|ne|lg)_(.*)$")


def mutate_text(text):
    """the mutant of one instruction's text, or None"""
    parts = text.split(None, 1)
    mn = parts[0]
    suffix = ""
    m = G.SUFFIX.search(mn)
    if m:
        suffix, mn = m.group(0), mn[:m.start()]
    new = SWAPS.get(mn)
    if new is None:
        c = CMP_RE.match(mn)
        if c:
            new = "%s_%s_%s" % (c.group(1), CMP[c.group(2)], c.group(3))
    if new is None:
        return None
    return new + suffix + (" " + parts[1] if len(parts) > 1 else "")


def mutants(asm_path, symbol, executed_lines, n, seed):
    """n mutants [(program index, line, old text, new text)] drawn from the instructions of `symbol` the captures executed"""
    prog, _, _ = G.parse_file(asm_path)
    a, b = G.function_extent(asm_path, symbol)
    ex = set(executed_lines)
    cands = []
    for i in range(a, b):
        if prog[i].line in ex:
            t = mutate_text(prog[i].text)
            if t is not None:
                cands.append((i, prog[i].line, prog[i].text, t))
    rng = random.Random(seed)
    rng.shuffle(cands)
    return cands[:n], len(cands)


def _one(job):
    asm_path, (idx, line, old, new), replays, cap = job
    os.environ.pop("SNK_ISA_COV_DIR", None)             # (a mutant's replays are no evidence of what the shipped code executes)
    prog, _, _ = G.parse_file(asm_path)
    keep = prog[idx]
    verdict = None
    try:
        try:
            prog[idx] = G.make_ins(new, line)
            if prog[idx].fn is G.ex_unknown:
                return (line, old, new, "not a mutant: no semantics for the swapped opcode", None)
        except Exception as ex:
            return (line, old, new, "not a mutant: %r" % (ex,), None)
        for dump, k in replays:
            try:
                info, diffs = G.replay(dump, k, asm_path, verbose=False, garbage=1, max_wave_instructions=cap)
                if diffs or info["scalar_loads_of_words_written_in_this_launch"]:
                    verdict = "memory differs (%s launch %d)" % (os.path.basename(dump), k)
            except G.Hazard as ex:
                verdict = "hazard: %s" % (str(ex)[:120],)
            except Exception as ex:                      # an address outside every allocation, an operand the handler refuses, ...
                verdict = "fault: %s: %s" % (type(ex).__name__, str(ex)[:120])
            if verdict:
                break
    finally:
        prog[idx] = keep
    return (line, old, new, verdict or "SURVIVED", len(replays))


def run_mutants(asm_path, muts, replays_for_line, cap=400000, workers=None):
    """replays_for_line(line) -> [(dump dir, launch)] of the captures that execute that line, most specific first.
    -> [(line, old, new, verdict, replays tried)]"""
    jobs = [(asm_path, m, replays_for_line(m[1]), cap) for m in muts]
    with concurrent.futures.ProcessPoolExecutor(max_workers=workers or min(8, os.cpu_count() or 1)) as pool:
        return list(pool.map(_one, jobs))
