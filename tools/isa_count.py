"""Instruction mix of a line range of a gfx950 assembly file (the budget tables of profiles/r04_isa_budget.md):
    python tools/isa_count.py file.s first last [first last ...]
classes: valu (v_*), salu (s_* arithmetic / moves / compares), sopp (s_nop, s_waitcnt, s_setprio, s_set_gpr_idx_*: issued by the
scalar side but not ALU work), branch (s_cbranch*, s_branch), lds (ds_*), vmem (global_* / scratch_* / buffer_*), smem (s_load*)"""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_nop", "s_waitcnt", "s_setprio", "s_set_gpr_idx", "s_barrier", "s_sleep")):
        return "sopp"
    if op.startswith("s_"):
        return "salu"
    return "other"


def count(lines):
    c = {}
    for raw in lines:
        ins = raw.split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        k = classify(ins.split()[0])
        c[k] = c.get(k, 0) + 1
    return c


if __name__ == "__main__":
    txt = open(sys.argv[1]).read().splitlines()
    a = [int(x) for x in sys.argv[2:]]
    for lo, hi in zip(a[::2], a[1::2]):
        c = count(txt[lo - 1:hi])
        print(f"{lo}-{hi}: " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
