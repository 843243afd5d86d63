"""SGPR-spill traffic of the tiled kernel's instances, from the assembly the build keeps (VERDICT r4 #4a).

LLVM spills scalar registers into LANES of vector registers it sets aside: a spill is `v_writelane_b32 vS, sN, lane`, a reload
`v_readlane_b32 sN, vS, lane`.  A spill carrier is a VGPR that nothing but those two instructions ever touches (the kernel's own
cross-lane reads -- `rl()` / `wl()` of snk_common.hip.h -- go through registers that vector instructions also use).  Per kernel
instance: the carriers, the static count of spill stores / reloads, and WHERE they sit -- per basic block with the block's loop
depth (blocks between a label and the farthest backward branch to it), so that the reloads inside the phase-1 octet loop (executed
8 x per tile-mate) can be told from the ones executed once per tile or once per launch.

    python tools/isa_spills.py                   table for the two headline instances
    python tools/isa_spills.py --all             every instance
    python tools/isa_spills.py --json out.json
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_static  # noqa: E402


def kernels(path=isa_static.ASM):
    """{mangled name: [lines of the body]}"""
    out, cur = {}, None
    with open(path) as fh:
        for line in fh:
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
            if m:
                cur = m.group(1)
                out[cur] = []
                continue
            if cur is not None:
                out[cur].append(line.rstrip("\n"))
                if line.split(";")[0].strip().startswith("s_endpgm"):
                    cur = None
    return out


def vregs(args):
    r = set(int(x) for x in re.findall(r"\bv(\d+)\b", args))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", args):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return r


def analyse(body):
    ins = []                     # (line index, op, args, label or None)
    labels = {}
    for i, raw in enumerate(body):
        t = raw.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(ins)
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        ins.append((i, op, t[len(op):]))
    touched_other, wl, rd = set(), {}, {}
    for k, (i, op, args) in enumerate(ins):
        if op == "v_writelane_b32":
            v = int(re.match(r"\s*v(\d+)", args).group(1))
            wl.setdefault(v, []).append(k)
        elif op == "v_readlane_b32":
            v = int(re.search(r",\s*v(\d+)", args).group(1))
            rd.setdefault(v, []).append(k)
        else:
            touched_other |= vregs(args)
    carriers = sorted(v for v in set(wl) | set(rd) if v not in touched_other)
    # loop depth of every instruction: a backward branch to label L at instruction b makes [L, b] a loop
    depth = [0] * len(ins)
    loops = []
    for k, (i, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = args.strip()
            if tgt in labels and labels[tgt] <= k:
                loops.append((labels[tgt], k))
    for lo, hi in loops:
        for k in range(lo, hi + 1):
            depth[k] += 1
    by_depth = {}
    for v in carriers:
        for kind, lst in (("store", wl.get(v, [])), ("reload", rd.get(v, []))):
            for k in lst:
                d = by_depth.setdefault(depth[k], {"store": 0, "reload": 0})
                d[kind] += 1
    valu = sum(1 for _, op, _ in ins if op.startswith("v_"))
    all_wl = sum(len(x) for x in wl.values())
    all_rd = sum(len(x) for x in rd.values())
    sp_wl = sum(len(wl.get(v, [])) for v in carriers)
    sp_rd = sum(len(rd.get(v, [])) for v in carriers)
    # the hottest loop: the phase-1 octet loop of whole tiles of full-length reads = the SMALLEST loop that holds the LDS reads of
    # eight rows (the three shapes of phase 1 are three copies; any loop around them holds those reads too)
    hot = None
    for lo, hi in loops:
        n_lds = sum(1 for k in range(lo, hi + 1) if ins[k][1].startswith("ds_read"))
        if n_lds >= 24 and (hot is None or hi - lo < hot[1] - hot[0]):
            hot = (lo, hi, n_lds)
    hot_sp = {"store": 0, "reload": 0, "valu": 0}
    if hot:
        lo, hi, _ = hot
        hot_sp["valu"] = sum(1 for k in range(lo, hi + 1) if ins[k][1].startswith("v_"))
        for v in carriers:
            hot_sp["store"] += sum(1 for k in wl.get(v, []) if lo <= k <= hi)
            hot_sp["reload"] += sum(1 for k in rd.get(v, []) if lo <= k <= hi)
    return {"carriers": len(carriers), "valu_static": valu, "writelane_all": all_wl, "readlane_all": all_rd, "spill_stores": sp_wl, "spill_reloads": sp_rd,
            "own_readlane": all_rd - sp_rd, "own_writelane": all_wl - sp_wl,
            "by_loop_depth": {str(k): v for k, v in sorted(by_depth.items())}, "octet_loop": hot_sp}


def main():
    ks = kernels()
    names = list(ks)
    pretty = isa_static.demangle(names)
    res = {}
    for n, p in zip(names, pretty):
        if "snk_tiled_kernel" not in p:
            continue
        if "--all" not in sys.argv and p not in isa_static.HEADLINE:
            continue
        res[p] = analyse(ks[n])
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(res, fh, indent=1)
    for p, r in res.items():
        print(p)
        print(f"  static VALU {r['valu_static']}; v_readlane {r['readlane_all']} (spill reloads {r['spill_reloads']}, the kernel's own {r['own_readlane']}); "
              f"v_writelane {r['writelane_all']} (spill stores {r['spill_stores']}, own {r['own_writelane']}); carrier VGPRs {r['carriers']}")
        print("  by loop depth (0 = executed once per launch ... deeper = more often): " +
              "; ".join(f"depth {d}: {v['store']} stores / {v['reload']} reloads" for d, v in r["by_loop_depth"].items()))
        print(f"  phase-1 octet loop (8 reads, the full-length shape): {r['octet_loop']['valu']} VALU of which spill stores {r['octet_loop']['store']} / reloads {r['octet_loop']['reload']}")


if __name__ == "__main__":
    main()
