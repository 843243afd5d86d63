"""Instruction budget of the tiled kernel per READ from the assembly alone (what can be said about the kernel's
instruction count while no GPU is open).

    python tools/isa_budget.py [--baseline old.s --baseline-src dir]      builds snk_tiled.hip with line tables (-gline-tables-only:
                                   the same code, 80 s) and prints the table of the two headline instances
    python tools/isa_budget.py --asm file.s [--src dir] ...               an assembly made that way earlier (its sources in dir)
    python tools/isa_budget.py --json out.json

What it does.  Every instruction of a kernel instance is attributed to a REGION of the source (the .loc directives: phases of
snk_tiled.hip by marker lines, the adapter search by its header file; small inlined helpers inherit the region around them).  The
phase-1 octet loop is found structurally (the innermost loops with a tile-mate's 40 LDS reads; the three shapes of phase 1 are three
copies, the benchmark's whole tiles of full-length reads run the smallest) and is exact: its VALU count / 8 is the per-read figure.
For the other regions the static count is turned into a per-read figure with the region's ratio (instructions executed per read) /
(instructions in the code) of the build whose counters were MEASURED (--baseline: commit d12b8ba, profiles/r04_isa_budget.md section 2
gives the per-read figures of its regions, profiles/r04_c2_pmc.json the total of 56.83): the assumption is that a change removes
executed and not-executed instructions of a region alike -- fair for what this round removed (spill reloads, copies, address
arithmetic spread over the code), not a measurement.  The FULL instance has no hand-made table: its total is scaled as a whole.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_static  # noqa: E402
from isa_count import classify  # noqa: E402

# source regions of snk_tiled.hip (the first line that matches starts the region); other files inherit the region of the code around them
REGIONS = [
    ("phase 1", r"// -+ phase 1$"),
    ("hand-over + planes", r"// -+ hand-over: lane = 4 positions"),
    ("fix-up pass", r"// fix-up pass \(rare\)"),
    ("hand-over + planes", r"int nlowq = 0;"),
    ("adapter search", r"const bool good = lanev && !estat;"),
    ("pair level + phase 3", r"if \(ada_pos >= 0\) \{ R.inc_ada = 1;"),
    ("flush / drain (per launch)", r"^snk_tiled_kernel\("),
]
FILE_REGION = {"snk_adapter_bits.hip.h": "adapter search", "snk_adapter_bits.cuh": "adapter search"}
# VALU per read of the measured build's regions (profiles/r04_isa_budget.md section 2: static counts x trip counts of that build,
# divergent code counted as taken: sum 62.1 against 56.83 measured)
MEASURED_BUILD = {"phase 1 set-up": 1.5, "phase-1 octet loop": 18.6, "hand-over + planes": 12.9, "fix-up pass": 1.4, "adapter search: the rest": 9.2,
                  "adapter search: screen loops": 6.0, "pair level + phase 3": 12.5}
MEASURED_VALU = {False: 56.83, True: 85.4}       # PMC, per read: configs[1] / configs[2] parameters (profiles/r04_c2_pmc.json, r04_c3_pmc.json)


def build_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="snk_budget_"), "snk_tiled_g.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "--cuda-device-only", "-S",
                           "snk_tiled.hip", "-o", out], cwd=os.path.join(ROOT, "soapnuke_amd", "csrc"), stderr=subprocess.DEVNULL)
    return out


def parse_kernels(path):
    files, out, cur = {}, {}, None
    with open(path) as fh:
        for line in fh:
            m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
            if m:
                files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
                continue
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
                out[cur] = []
                continue
            if cur is not None:
                out[cur].append(line.rstrip("\n"))
                if line.split(";")[0].strip().startswith("s_endpgm"):
                    cur = None
    return files, out


def analyse(body, files, src_dir):
    tiled_name = [f for f in files.values() if f.startswith("snk_tiled")][0]
    tl = open(os.path.join(src_dir, tiled_name), errors="replace").read().splitlines()
    starts = []
    for name, pat in REGIONS:
        for i, text in enumerate(tl, 1):
            if re.search(pat, text):
                starts.append((i, name))
                break
    starts.sort()

    def region_of(line):
        r = "phase 1"                                # (the top of process_tile: masks and addresses of phase 1)
        for i, name in starts:
            if line >= i:
                r = name
        return r

    ins, labels = [], {}
    loc, cur_region = (tiled_name, 0), "flush / drain (per launch)"
    for raw in body:
        t = raw.split(";")[0].strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not t:
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(ins)
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        if loc[0] == tiled_name and loc[1] > 0:
            cur_region = region_of(loc[1])
        ins.append({"op": op, "cls": classify(op), "region": FILE_REGION.get(loc[0], cur_region), "args": t[len(op):]})
    merged = {}
    for k, x in enumerate(ins):
        if x["op"].startswith(("s_cbranch", "s_branch")):
            tgt = x["args"].strip()
            if tgt in labels and labels[tgt] <= k:
                merged[labels[tgt]] = max(merged.get(labels[tgt], k), k)
    loops = sorted(merged.items())
    # the octet loops: innermost loops with the LDS reads of eight rows (2 + NS reads each)
    octets = []
    for lo, hi in loops:
        nrd = sum(1 for k in range(lo, hi + 1) if ins[k]["op"].startswith("ds_read"))
        if nrd >= 24 and not any(a > lo and b <= hi or a >= lo and b < hi for a, b in loops):
            octets.append((lo, hi, nrd))
    in_octet = [False] * len(ins)
    for lo, hi, _ in octets:
        for k in range(lo, hi + 1):
            in_octet[k] = True
    # the screen loops of the adapter search (snk_adapter_bits: one funnel shift per plane word and adapter character): innermost,
    # small, made of v_alignbit + v_bitop3; a body that holds two shifts per word walks two characters per trip
    screens, in_screen = [], [False] * len(ins)
    for lo, hi in loops:
        if hi - lo > 160 or any((a > lo and b <= hi) or (a >= lo and b < hi) for a, b in loops):
            continue
        seg = ins[lo:hi + 1]
        nal = sum(1 for x in seg if x["op"].startswith("v_alignbit"))
        nv = sum(1 for x in seg if x["cls"] == "valu")
        if nal >= 4 and all(x["region"] == "adapter search" for x in seg if x["cls"] == "valu"):
            screens.append({"valu": nv, "alignbit": nal, "chars": 2 if nal >= 8 else 1})
            for k in range(lo, hi + 1):
                in_screen[k] = True
    regions = {}
    for k, x in enumerate(ins):
        r = "phase-1 octet loop (all copies)" if in_octet[k] else ("phase 1 set-up" if x["region"] == "phase 1" else x["region"])
        if in_screen[k]:
            r = "adapter search: screen loops"
        elif r == "adapter search":
            r = "adapter search: the rest"
        d = regions.setdefault(r, {})
        d[x["cls"]] = d.get(x["cls"], 0) + 1
    best = min(octets, key=lambda o: o[1] - o[0]) if octets else None
    octet = None
    if best:
        seg = ins[best[0]:best[1] + 1]
        octet = {"instructions": len(seg), "ds_read": best[2]}
        for x in seg:
            octet[x["cls"]] = octet.get(x["cls"], 0) + 1
    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if v else 0
    two, one = med([x["valu"] for x in screens if x["chars"] == 2]), med([x["valu"] for x in screens if x["chars"] == 1])
    # 17.5 screened characters per tile-mate and adapter (S - 1 = 15 / 20 for the two README adapters), spread over the letter loops
    # of the two 32-bit halves of the character masks: about 85 % of them go through the two-character body where there is one
    steps = 17.5
    screen_per_read = (steps * 0.85 * two / 2.0 + steps * 0.15 * one) / 64.0 if two else steps * one / 64.0
    return {"regions": regions, "octet": octet, "octet_copies": len(octets), "static_valu": sum(1 for x in ins if x["cls"] == "valu"),
            "screen": {"loops": len(screens), "valu_two_characters": two, "valu_one_character": one, "per_read": screen_per_read}}


def pick(path, src_dir, want_all=False):
    files, ks = parse_kernels(path)
    names = list(ks)
    pretty = isa_static.demangle(names)
    res = {}
    for n, p in zip(names, pretty):
        if "snk_tiled_kernel" not in p or not re.search(r"snk_tiled_kernel<5, ?(true|false), ?true, ?16", p):
            continue
        if "TileShape<160" not in p and any("TileShape<160" in q and "snk_tiled_kernel<5" in q for q in pretty):
            continue
        full = p.split(",")[1].strip() == "true"
        res[full] = (p, analyse(ks[n], files, src_dir))
    return res


def main():
    args = sys.argv[1:]
    asm = args[args.index("--asm") + 1] if "--asm" in args else build_asm()
    src_dir = args[args.index("--src") + 1] if "--src" in args else os.path.join(ROOT, "soapnuke_amd", "csrc")
    new = pick(asm, src_dir)
    old = pick(args[args.index("--baseline") + 1], args[args.index("--baseline-src") + 1]) if "--baseline" in args else {}
    out = {}
    for full in (False, True):
        if full not in new:
            continue
        p, r = new[full]
        print(p)
        o = old.get(full, (None, None))[1]
        rows, total_new, total_old_model = [], 0.0, 0.0
        oc = r["octet"]["valu"] / 8.0
        print(f"  phase-1 octet loop (the full-length shape, of {r['octet_copies']} copies): {r['octet']}  -> {oc:.2f} VALU per read (exact)"
              + (f"; measured build: {o['octet']['valu'] / 8.0:.2f}" if o else ""))
        total_new += oc
        if o:
            total_old_model += o["octet"]["valu"] / 8.0
        print("  region                                   static VALU   measured build: static   per read (its table)   -> per read here")
        sp = r["screen"]["per_read"]
        print(f"  adapter screen loops: {r['screen']}  -> {sp:.2f} VALU per read (structural)" + (f"; measured build: {o['screen']['per_read']:.2f}" if o else ""))
        total_new += sp
        if o:
            total_old_model += o["screen"]["per_read"]
        for name in ("phase 1 set-up", "hand-over + planes", "fix-up pass", "adapter search: the rest", "pair level + phase 3"):
            sv = r["regions"].get(name, {}).get("valu", 0)
            if o:
                ov = o["regions"].get(name, {}).get("valu", 0)
                # (configs[2] parameters: no hand-made table -- the configs[1] table's ratios per region)
                per_old = MEASURED_BUILD[name] if not full else MEASURED_BUILD[name] * (ov / max(1, old[False][1]["regions"].get(name, {}).get("valu", 1)) if False in old else 1.0)
                per_new = per_old * sv / max(1, ov)
                total_new += per_new
                total_old_model += per_old
                print(f"  {name:38s} {sv:10d}   {ov:20d}   {per_old:20.2f}   {per_new:14.2f}")
                rows.append({"region": name, "static_valu": sv, "measured_build_static_valu": ov, "measured_build_per_read": per_old, "per_read": per_new})
            else:
                print(f"  {name:38s} {sv:10d}")
                rows.append({"region": name, "static_valu": sv})
        res = {"kernel": p, "static_valu": r["static_valu"], "octet_loop": r["octet"], "octet_valu_per_read": oc, "screen_loops": r["screen"], "regions": rows,
               "flush_static_valu": r["regions"].get("flush / drain (per launch)", {}).get("valu", 0)}
        if o:
            scale = MEASURED_VALU[full] / total_old_model
            res.update({"model_valu_per_read": total_new, "measured_build_model": total_old_model, "measured_build_pmc": MEASURED_VALU[full],
                        "projected_valu_per_read": total_new * scale})
            print(f"  model: {total_new:.1f} VALU per read here, {total_old_model:.1f} for the measured build, whose counters say {MEASURED_VALU[full]:.2f}"
                  f"  ->  projected {total_new * scale:.1f} VALU per read  (static VALU of the instance: {r['static_valu']}, measured build {o['static_valu']})")
        out[p] = res
    if "--json" in args:
        with open(args[args.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
