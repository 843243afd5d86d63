"""Lint of the tiled kernel's generated code (gfx950 assembly, `hipcc -S`): phase 1 reads its rows from LDS with hand-placed asm
`ds_read` instructions and counted `s_waitcnt lgkmcnt(N)` waits (snk_tiled.hip) -- to the compiler the outputs of such an asm
statement exist from the moment it is issued, so a register copy on a loop edge or a spill placed ahead of the wait moves bits
the LDS unit has not delivered yet (round 4 met exactly that: an intermittent, one-read-per-million difference in one build).

The check walks every `snk_tiled_kernel` instance in text order with the in-order LDS return queue the hardware keeps
(LDS instructions return in issue order; `s_waitcnt lgkmcnt(N)` = all but the newest N have returned) and reports every
instruction that names a VGPR whose `ds_read` has not been covered by a wait yet.  Control flow is ignored (the phase-1 loops are
straight-line apart from the DMA issue blocks, which touch no row register); loop back-edges are covered by walking each
function twice with the queue carried over.

    python tools/isa_lint.py file.s [...]          exit status 1 when something is reported
    python tools/isa_lint.py --build               compiles soapnuke_amd/csrc/snk_tiled.hip to assembly first (about 100 s)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_function(name, lines, report, passes=2):
    queue = []          # outstanding LDS instructions, oldest first: (set of destination VGPRs, line number, text)
    found = 0
    for walk in range(passes):
        for no, raw in lines:
            ins = raw.split(";")[0].strip()
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            op = ins.split()[0]
            if op == "s_waitcnt":
                m = LGKM.search(ins)
                if m:
                    # at most `keep` of the outstanding LDS + SMEM operations may still be in flight.  LDS operations return in
                    # order; scalar loads do NOT, so in the worst case the completed ones are all the scalar loads first: of the
                    # LDS operations only (completed - scalar loads in flight) oldest ones are certain (a compiler-made s_load
                    # between an asm ds_read and its counted wait would satisfy the wait in its place)
                    keep = int(m.group(1))
                    if keep == 0:
                        queue = []
                    else:
                        done = len(queue) - keep
                        n_smem = sum(1 for q in queue if q[3] == "smem")
                        sure = max(0, done - n_smem)
                        out = []
                        for q in queue:
                            if q[3] == "lds" and sure > 0:
                                sure -= 1
                                continue
                            out.append(q)
                        queue = out
                elif "lgkmcnt" not in ins and re.fullmatch(r"s_waitcnt\s+\S+", ins) and "vmcnt" not in ins and "expcnt" not in ins:
                    queue = []      # a numeric immediate: treat as a full wait
                continue
            if op in ("s_endpgm",):
                queue = []
                continue
            used = vregs(ins[len(op):])
            pending = set().union(*[q[0] for q in queue if q[3] == "lds"]) if queue else set()
            hit = used & pending
            if hit and walk == passes - 1 or (hit and walk == 0 and passes == 1):
                src = [q for q in queue if q[3] == "lds" and q[0] & hit][0]
                report.append(f"{name}: line {no}: `{ins}` touches v{sorted(hit)} before the wait that covers `{src[2]}` (line {src[1]})")
                found += 1
            if op.startswith("ds_"):
                ops = [x.strip() for x in ins[len(op):].split(",")]
                dest = set()
                returns = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in op
                if returns and ops:
                    dest = vregs(ops[0])
                queue.append((dest, no, ins, "lds"))
            elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_dcache") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
                queue.append((set(), no, ins, "smem"))      # shares the counter and returns out of order (see s_waitcnt above)
    return found


def lint_file(path):
    report = []
    cur, body = None, []
    funcs = 0
    with open(path) as fh:
        for no, line in enumerate(fh, 1):
            m = re.match(r"^(_Z\w*snk_tiled_kernel\w*):", line)
            if m:
                cur, body = m.group(1), []
                continue
            if cur is not None:
                body.append((no, line))
                if "s_endpgm" in line:
                    funcs += 1
                    lint_function(cur[:90], body, report)
                    cur = None
    return funcs, report


def main():
    args = sys.argv[1:]
    files = [a for a in args if not a.startswith("--")]
    if "--build" in args:
        out = os.path.join(tempfile.mkdtemp(prefix="snk_isa_"), "snk_tiled.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out, "snk_tiled.hip"],
                              cwd=os.path.join(ROOT, "soapnuke_amd", "csrc"), stderr=subprocess.DEVNULL)
        files.append(out)
    bad = 0
    for f in files:
        funcs, report = lint_file(f)
        print(f"{f}: {funcs} snk_tiled_kernel instances, {len(report)} finding(s)")
        for r in report[:40]:
            print("  " + r)
        bad += len(report)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
