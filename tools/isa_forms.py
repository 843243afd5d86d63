"""tools/isa_forms.py <proven assembly dir> [<build dir>] : the instruction FORMS of the shipped build that no hardware-run build contained.

A form = mnemonic (with its encoding suffix) + the kinds of its operands (v / s / literal / inline constant / vcc / exec / m0 ...) + its
modifiers (DPP control family, row / bank masks other than 0xf, bound_ctrl, SDWA selects, op_sel, clamp, offset present, glc / slc ...).
The emulated tier and the instruction tier (tools/gfx950_interp.py) vouch for what a form DOES only as far as this repository's reading
of the ISA manual goes; a form that a hardware-green build already contained has also been executed by silicon with the results the
oracle expects.  This lists the rest: forms of soapnuke_amd/csrc/build/*.s that occur nowhere in the assembly of the last build that
ran on an MI355X (abl/hw_d12b8ba, compiled with `hipcc -S --offload-device-only` into <proven assembly dir>), per source file, with
counts and one example line each -- the residual "never on silicon in this project" list a first contact should look at first."""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gfx950_interp as G          # noqa: E402

DPP = ("quad_perm", "row_shl", "row_shr", "row_ror", "wave_shl", "wave_shr", "wave_rol", "wave_ror", "row_newbcast", "row_bcast", "row_mirror", "row_half_mirror")


def kind(op):
    k = op.kind
    if k in ("v", "s", "a"):
        return k + (str(op.cnt) if op.cnt > 1 else "")
    return k


def form(ins):
    mods = []
    for k, v in sorted(ins.mods.items()):
        if k in ("row_mask", "bank_mask"):
            if str(v).lower() not in ("0xf", "15"):
                mods.append(k + "!=f")
        elif k in ("offset", "offset0", "offset1"):
            mods.append("offset")
        elif k.endswith("_sel") or k == "dst_unused":
            mods.append(k + "=" + str(v))
        else:
            mods.append(k)
    mods += sorted(ins.flags)
    return " ".join([ins.mn, ",".join(kind(o) for o in ins.ops)] + mods)


def forms_of(path):
    out = {}
    prog, _, _ = G.parse_file(path)
    for ins in prog:
        e = out.setdefault(form(ins), [0, ins.text, ins.line])
        e[0] += 1
    return out


def main():
    proven_dir = sys.argv[1]
    build = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(HERE), "soapnuke_amd", "csrc", "build")
    proven = {}
    for f in sorted(glob.glob(os.path.join(proven_dir, "*.s"))):
        for k, v in forms_of(f).items():
            proven[k] = proven.get(k, 0) + v[0]
    print("%d instruction forms in the hardware-run build's assembly (%s)" % (len(proven), proven_dir))
    total_new = 0
    for f in sorted(glob.glob(os.path.join(build, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        mine = forms_of(f)
        new = {k: v for k, v in mine.items() if k not in proven}
        print("\n%s: %d forms, %d of them in no hardware-run build" % (os.path.basename(f).split("-hip-")[0], len(mine), len(new)))
        for k, (n, text, line) in sorted(new.items(), key=lambda kv: -kv[1][0]):
            print("  %6d x  %-60s  e.g. line %d: %s" % (n, k[:60], line, text[:72]))
        total_new += len(new)
    print("\n%d forms in all that only this repository's reading of the manual vouches for" % total_new)


if __name__ == "__main__":
    main()
