"""tools/isa_allocas.py [source.hip ...]: what the kernels keep in scratch memory and reach through generic pointers, without a GPU.

For every HIP source of soapnuke_amd/csrc (default: all of build.SOURCES that are .hip) the device code is compiled with
-save-temps, the bitcode optimised with `opt -O3`, and per kernel are listed: the private objects (`alloca`) that survive
optimisation -- an object whose address is compared, or that is indexed with a run-time value, stays in scratch memory --, and from
the assembly the numbers of flat / scratch / scalar-load instructions, the VGPRs and the spilled VGPRs.  Round 4 found three such
objects this way (the Coop of the device inflate, the ReadState pair of the long-read decide kernel, the contaminant planes handed
to functions the compiler had not inlined); profiles/r04_allocas.txt is the committed output."""
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "soapnuke_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def one(src, tmp):
    name = os.path.splitext(os.path.basename(src))[0]
    d = os.path.join(tmp, name)
    os.makedirs(d)
    kept = os.path.join(CSRC, "build")           # soapnuke_amd/build.py keeps bitcode and assembly of the shipped objects (same flags)
    kbc = os.path.join(kept, name + "-hip-amdgcn-amd-amdhsa-gfx950.bc")
    if os.path.dirname(os.path.abspath(src)) == CSRC and os.path.exists(kbc) and os.path.exists(kbc[:-3] + ".s") and not os.environ.get("ISA_ALLOCAS_RECOMPILE"):
        import glob
        deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
        if all(os.path.getmtime(f) <= os.path.getmtime(kbc) for f in deps):
            d = kept
    r = subprocess.CompletedProcess([], 0) if d == kept else subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "x.o", "-save-temps=obj"],
                       cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        return name, "compile failed:\n" + r.stdout[-2000:]
    bc = os.path.join(d, name + "-hip-amdgcn-amd-amdhsa-gfx950.bc")
    asm = os.path.join(d, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    ll = subprocess.run([os.path.join(LLVM, "opt"), "-O3", bc, "-S", "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    allocas, cur = {}, None
    for line in ll.split("\n"):
        m = re.match(r"define .*@([A-Za-z0-9_]+)\(", line)
        if m:
            cur = m.group(1)
        m = re.search(r"= alloca (.*?), align", line)
        if m and cur and "kernel" in cur:
            allocas.setdefault(cur, []).append(m.group(1))
    counts, meta, cur = {}, {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            counts[cur] = {"flat": 0, "scratch": 0, "s_load": 0}
        t = line.strip().split(" ")[0] if cur else ""
        if t.startswith("flat_"):
            counts[cur]["flat"] += 1
        elif t.startswith("scratch_"):
            counts[cur]["scratch"] += 1
        elif t.startswith("s_load"):
            counts[cur]["s_load"] += 1
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            meta_cur = m.group(1)
            meta[meta_cur] = {}
        for key in (".vgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size"):
            m = re.match(r"\s+\%s:\s+(\d+)" % key, line)
            if m and meta:
                meta[meta_cur][key] = int(m.group(1))
    out = []
    for k in sorted(meta):
        c = counts.get(k, {"flat": 0, "scratch": 0, "s_load": 0})
        try:
            short = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
        except OSError:
            short = k
        out.append(f"  {short}: vgprs {meta[k].get('.vgpr_count')}, spilled {meta[k].get('.vgpr_spill_count')}, scratch bytes {meta[k].get('.private_segment_fixed_size')}, "
                   f"flat {c['flat']}, scratch instr {c['scratch']}, s_load {c['s_load']}" + ("; private objects: " + ", ".join(allocas[k]) if k in allocas else ""))
    extra = [f"  (function not inlined) {k}: flat {v['flat']}, scratch instr {v['scratch']}" for k, v in sorted(counts.items()) if k not in meta and (v["flat"] or v["scratch"])]
    return name, "\n".join(out + extra)


def main():
    srcs = [os.path.join(CSRC, s) if not os.path.isabs(s) else s for s in sys.argv[1:]]
    if not srcs:
        sys.path.insert(0, ROOT)
        from soapnuke_amd import build
        srcs = [os.path.join(CSRC, s) for s in build.SOURCES if s.endswith(".hip")]
    with tempfile.TemporaryDirectory(prefix="snk_allocas_") as tmp, cf.ThreadPoolExecutor(max_workers=4) as ex:
        for name, text in ex.map(lambda s: one(s, tmp), srcs):
            print(f"== {name}")
            print(text)


if __name__ == "__main__":
    main()
