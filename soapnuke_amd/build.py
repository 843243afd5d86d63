"""Builds the gfx950 shared library in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsnk_filter.so")
SOURCES = ["snk_filter.cpp", "snk_generic.hip", "snk_tiled.hip", "snk_rmdup.hip", "snk_contam.hip", "snk_long.hip", "snk_fastq.hip", "snk_gzip.hip", "snk_inflate.hip"]
def _headers():
    """every header a kernel source can include: csrc/*.hip.h|*.h and include/*.h (a missing entry once let a stale
    library survive a plane-store layout change)"""
    import glob
    inc = os.path.join(HERE, "..", "include")
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(inc, "*.h")))


HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, f) for f in SOURCES] + _headers())


OBJ = os.path.join(CSRC, "build")            # objects + every source's gfx950 assembly (git-ignored, not shipped to the GPU box)
TILED_ASM = os.path.join(OBJ, "snk_tiled-hip-amdgcn-amd-amdhsa-gfx950.s")


def build(force=False, verbose=False, lint=True):
    """every source to its own object (in parallel), one link; the device assembly is kept (-save-temps: tools/isa_*.py read it,
    tools/gfx950_interp.py runs it); the tiled kernel's is checked by tools/isa_lint.py: no use of an LDS row register ahead of
    the s_waitcnt that covers it"""
    if not force and not needs_build():
        return LIB
    import concurrent.futures as cf
    os.makedirs(OBJ, exist_ok=True)
    flags = [f for f in FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".", "_") + ".o")
        cmd = [HIPCC] + flags + ["-c", os.path.join("..", src), "-o", obj] + ["-save-temps=obj"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, cwd=OBJ, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{src}:\n{r.stdout[-4000:]}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=OBJ)
    if lint and os.path.exists(TILED_ASM):
        import importlib.util
        spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(HERE, "..", "tools", "isa_lint.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        funcs, report = mod.lint_file(TILED_ASM)
        if verbose:
            print(f"isa_lint: {funcs} snk_tiled_kernel instances, {len(report)} finding(s)")
        if funcs == 0 or report:
            os.unlink(LIB)
            raise RuntimeError("tools/isa_lint.py: " + ("no kernel found in " + TILED_ASM if funcs == 0 else "\n".join(report[:20])))
    return LIB


HOST = os.path.join(HERE, "host")
CLI = os.path.join(HERE, "SOAPnuke")
REPORT_LIB = os.path.join(HERE, "libsnk_report.so")
WIRE_TEST = os.path.join(HERE, "snk_wire_selftest")


def build_host(force=False, verbose=False):
    """The C++ host side: report writer library (plain g++) and the `SOAPnuke filter` CLI (links the
    C-ABI library with an $ORIGIN rpath)."""
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith((".cpp", ".h"))] + \
           [f for f in _headers() if os.sep + "include" + os.sep in f]
    newest = max(os.path.getmtime(f) for f in srcs + [LIB])
    if force or not os.path.exists(REPORT_LIB) or os.path.getmtime(REPORT_LIB) < newest:
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", REPORT_LIB, "snk_report.cpp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=HOST)
    if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < newest:
        # -march=x86-64-v3 (AVX2, BMI2: every host an MI355X sits in has them; main() checks and says so otherwise): the host inflate
        # core -- the larger half of the CLI's CPU time on .gz input -- decodes 371 instead of 323 MB/s per thread with it
        # (tools/micro/inflate_test.cpp, gzip -1 FASTQ, this container)
        cmd = [HIPCC, "-O2", "-march=x86-64-v3", "-std=c++17", "-o", CLI, "snk_main.cpp", "snk_report.cpp", "-L" + HERE, "-lsnk_filter", "-lz", "-pthread",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=HOST)
    if force or not os.path.exists(WIRE_TEST) or os.path.getmtime(WIRE_TEST) < newest:
        # the wires of the sharded run at world size 1 (test infrastructure: tests/test_wire_gpu.py)
        cmd = [HIPCC, "-O2", "-std=c++17", "-o", WIRE_TEST, "snk_wire_selftest.cpp", "-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=HOST)
    return CLI


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_host(force=True, verbose=True)
