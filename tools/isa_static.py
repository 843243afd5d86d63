"""Static figures of every kernel of the tiled source, read from the gfx950 assembly the build keeps
(soapnuke_amd/csrc/build/snk_tiled-hip-amdgcn-amd-amdhsa-gfx950.s): registers, spills, scratch and LDS bytes from the code-object
metadata, and the instruction mix of the kernel body (VALU / SALU / LDS / VMEM / branches / s_waitcnt / s_nop).  They need no GPU:
a change that moves the headline instances (`snk_tiled_kernel<5, *, true, 16, TileShape<160, 768, 4>>`: BASELINE configs[1] / [2]) shows
here before it is ever timed.

    python tools/isa_static.py                      table of all instances
    python tools/isa_static.py --json out.json      the same as JSON (profiles/r04_isa_static.json is such a file)
    python tools/isa_static.py --check base.json    exit 1 if a headline instance spills more, uses more scratch or has grown by
                                                    more than 2 % in instructions against the file (tests/test_isa_lint.py runs this)
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "soapnuke_amd", "csrc", "build", "snk_tiled-hip-amdgcn-amd-amdhsa-gfx950.s")
META = ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count")
HEADLINE = ("snk_tiled_kernel<5, false, true, 16, (anonymous namespace)::TileShape<160, 768, 4>>",
            "snk_tiled_kernel<5, true, true, 16, (anonymous namespace)::TileShape<160, 768, 4>>")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"> >", ">>", re.sub(r"(?<=[\w>])\(.*$", "", re.sub(r"^void \(anonymous namespace\)::", "", x))) for x in out]


def classify(op):
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")):
        return "branch"
    if op == "s_waitcnt":
        return "waitcnt"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def collect(path=ASM):
    meta, body, cur, entry = {}, {}, None, None
    with open(path) as fh:
        for line in fh:
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
            if m:
                cur = m.group(1)
                body[cur] = {}
                continue
            if cur is not None:
                ins = line.split(";")[0].strip()
                if ins and not ins.endswith(":") and not ins.startswith("."):
                    k = classify(ins.split()[0])
                    body[cur][k] = body[cur].get(k, 0) + 1
                    if ins.startswith("s_endpgm"):
                        cur = None
                continue
            if line.startswith("  - .agpr_count:"):            # a kernel's entry of amdhsa.kernels
                entry = {}
                continue
            if entry is not None:
                m = re.match(r"^    \.(\w+):\s+(\S+)\s*$", line)
                if m and m.group(1) in META:
                    entry[m.group(1)] = int(m.group(2))
                elif m and m.group(1) == "name":
                    meta[m.group(2)] = entry
    names = [n for n in meta if n in body]
    pretty = demangle(names)
    return {p: dict(meta[n], **{"insts_" + k: v for k, v in sorted(body[n].items())}, insts_total=sum(body[n].values())) for n, p in zip(names, pretty)}


def main():
    args = sys.argv[1:]
    data = collect()
    if "--json" in args:
        with open(args[args.index("--json") + 1], "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
    if "--check" in args:
        base = json.load(open(args[args.index("--check") + 1]))
        bad = []
        for k in HEADLINE:
            if k not in data or k not in base:
                bad.append(f"{k}: missing")
                continue
            a, b = data[k], base[k]
            for f in ("vgpr_spill_count", "private_segment_fixed_size", "vgpr_count"):
                if a[f] > b[f]:
                    bad.append(f"{k}: {f} {b[f]} -> {a[f]}")
            # scalar registers spilled into VGPR lanes: every reload is a v_readlane in a VALU slot (round 5: 232 -> 78); a handful
            # more is allocation noise, the old level is a regression
            if a.get("sgpr_spill_count", 0) > max(120, b.get("sgpr_spill_count", 0) + 16):
                bad.append(f"{k}: sgpr_spill_count {b.get('sgpr_spill_count')} -> {a.get('sgpr_spill_count')}")
            if a.get("insts_valu", 0) > b.get("insts_valu", 0) * 1.02:
                bad.append(f"{k}: {b.get('insts_valu')} -> {a.get('insts_valu')} VALU instructions")
            if a["insts_total"] > b["insts_total"] * 1.02:
                bad.append(f"{k}: {b['insts_total']} -> {a['insts_total']} instructions")
        print("\n".join(bad) if bad else "isa_static: the headline instances are within the committed figures")
        return 1 if bad else 0
    cols = ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "insts_total", "insts_valu", "insts_salu", "insts_lds",
            "insts_vmem", "insts_branch", "insts_waitcnt", "insts_nop")
    print("\t".join(("kernel",) + cols))
    for k in sorted(data):
        print("\t".join([k] + [str(data[k].get(c, 0)) for c in cols]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
