"""Builds the SIMT-emulated twin of libsnk_filter.so: the same sources (soapnuke_amd/csrc), compiled for the host with
tests/simt/hip/hip_runtime.h standing in for the HIP runtime.  Test infrastructure only -- see hip/hip_runtime.h."""
import concurrent.futures as cf
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "soapnuke_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libsnk_filter_simt.so")
CXX = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")       # clang: the sources use ext_vector_type and address_space attributes
FLAGS = ["-std=c++17", "-O2", "-g1", "-gdwarf-4", "-fPIC", "-w", "-pthread", "-I" + HERE, "-I" + CSRC, "-fno-strict-aliasing", "-fno-omit-frame-pointer",
         "-fconvergent-functions", "-mllvm", "-disable-tail-duplicate", "-mllvm", "-disable-early-taildup", "-mllvm", "-tail-dup-placement=0"]      # every function may hold wave-level operations (what hipcc assumes for device code): no duplication of calls into the arms of lane-dependent branches -- in the middle end by the attribute, in the x86 backend, which does not know it, by switching tail duplication off


# SIMT_EXTRA_CXXFLAGS / SIMT_TAG: a variant of the emulated library (a -D switch of the kernels, e.g. -DSNK_PAIR=1) under its own name
if os.environ.get("SIMT_EXTRA_CXXFLAGS"):
    FLAGS = FLAGS + os.environ["SIMT_EXTRA_CXXFLAGS"].split()
    LIB = os.path.join(OUT, "libsnk_filter_simt_" + os.environ.get("SIMT_TAG", "variant") + ".so")


def sources():
    import sys
    sys.path.insert(0, ROOT)
    from soapnuke_amd import build
    return list(build.SOURCES), build._headers()


def needs_build(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    srcs, hdrs = sources()
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in srcs] + hdrs + [os.path.join(HERE, f) for f in ("simt_runtime.cpp", "simt_gfx950.h", "hip/hip_runtime.h")]
    return any(os.path.getmtime(f) > t for f in deps)


def build(force=False, extra=(), lib=None):
    lib = lib or LIB
    if not force and not needs_build(lib):
        return lib
    os.makedirs(OUT, exist_ok=True)
    srcs, _ = sources()
    tag = os.path.basename(lib).replace(".so", "")

    def one(src):
        path = os.path.join(CSRC, src) if src != "simt_runtime.cpp" else os.path.join(HERE, src)
        obj = os.path.join(OUT, tag + "_" + src.replace(".", "_") + ".o")
        r = subprocess.run([CXX] + FLAGS + list(extra) + ["-x", "c++", "-c", path, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{src}:\n{r.stdout[-4000:]}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, srcs + ["simt_runtime.cpp"]))
    subprocess.check_call([CXX, "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic", "-o", lib] + list(extra) + objs + ["-ldl"])
    return lib


CLI = os.path.join(OUT, "SOAPnuke_simt")


def build_cli_tsan(force=False):
    """the CLI's own code (readers, the writer, slot makers, the parallel and the device gunzip orchestration) with ThreadSanitizer,
    linked against the plain emulated library (whose fibers the sanitizer would not follow)"""
    return build_cli(force, extra=("-fsanitize=thread",), tag="_tsan", lib_tag="")


def build_cli(force=False, extra=(), tag="", lib_tag=None):
    """`SOAPnuke filter` (soapnuke_amd/host) linked against the emulated library: the whole host side -- readers, gzip, shards, rmdup
    orchestration, reports -- runs on the CPU and can be compared with the reference binary file by file"""
    lib_tag = tag if lib_tag is None else lib_tag
    lib = build(force, extra if lib_tag else (), os.path.join(OUT, "libsnk_filter_simt" + lib_tag + ".so"))
    cli = CLI + tag
    host = os.path.join(ROOT, "soapnuke_amd", "host")
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith((".cpp", ".h"))]
    if not force and os.path.exists(cli) and os.path.getmtime(cli) >= max(os.path.getmtime(f) for f in srcs + [lib]):
        return cli
    cmd = [CXX] + FLAGS + list(extra) + ["-x", "c++", os.path.join(host, "snk_main.cpp"), os.path.join(host, "snk_report.cpp"), "-x", "none",
                                         "-o", cli, "-L" + OUT, "-lsnk_filter_simt" + lib_tag, "-lz", "-ldl", "-Wl,-rpath," + OUT]
    subprocess.check_call(cmd)
    return cli


def build_wire_selftest(force=False):
    """soapnuke_amd/host/snk_wire_selftest.cpp against the emulated runtime: the host wire's checks run, the RCCL half says why it cannot"""
    lib = build(force)
    exe = os.path.join(OUT, "snk_wire_selftest_simt")
    host = os.path.join(ROOT, "soapnuke_amd", "host")
    srcs = [os.path.join(host, f) for f in ("snk_wire_selftest.cpp", "snk_wire.h")]
    if not force and os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(f) for f in srcs + [lib]):
        return exe
    subprocess.check_call([CXX] + FLAGS + ["-x", "c++", srcs[0], "-x", "none", "-o", exe, "-L" + OUT, "-lsnk_filter_simt", "-ldl", "-Wl,-rpath," + OUT])
    return exe


if __name__ == "__main__":
    print(build(force=True))
    print(build_cli(force=True))
