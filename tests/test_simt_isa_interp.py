"""The gfx950 INSTRUCTIONS of the tiled kernel -- the assembly hipcc leaves next to the object the product ships
(soapnuke_amd/csrc/build/*.s) -- run on the CPU by tools/gfx950_interp.py, on launches captured from the emulated library
(tests/simt: SIMT_DUMP_DIR), and compared byte for byte with the device memory the emulated C++ twin left behind (which the capture
has compared with the oracle).  No GPU.

What this adds to tests/test_simt_kernels.py (HIP sources compiled for the host): the compiler's output and the hand-placed gfx950
blocks themselves are executed -- ds_read / ds_add with immediate offsets behind counted s_waitcnt, the LDS DMA, DPP and permlane
transposes, v_writelane / v_readlane spills, scratch spills, s_set_gpr_idx -- under an adversarial completion model: every
asynchronous result stays poisoned until the s_waitcnt that covers it by the counters' rules (in-order vmcnt / LDS lgkmcnt, scalar
loads out of order), reading it earlier is a failure.  The negative controls at the end show the checker has teeth.
What it still is not: hardware.  Timing, cache behaviour and the ISA manual's own errata stay with the first run on an MI355X."""
import json
import os
import re
import subprocess
import sys

import pytest

import snk_testlib as T

sys.path.insert(0, os.path.join(T.ROOT, "tools"))
import gfx950_interp as G          # noqa: E402

ASM = os.path.join(T.ROOT, "soapnuke_amd", "csrc", "build", "snk_tiled-hip-amdgcn-amd-amdhsa-gfx950.s")
HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (capture spec, environment of the capture, part of the kernel instance's mangled name that must have run)
CASES = {
    "pe150_c2_static": (dict(case="C2_adatrim_lowq", n=1100, L=150), {}, "ILi5ELb0ELb1ELi16ENS_9TileShapeILi160"),
    "pe150_c3_full_ragged": (dict(case="C3_full", n=1100, L=150, var_len=True), {}, "ILi5ELb1ELb1ELi16ENS_9TileShapeILi160"),
    "pe250_c3_full_ragged": (dict(case="C3_full", n=1100, L=250, var_len=True), {}, "ILi8ELb1ELb1ELi16ENS_9TileShapeILi256"),
    "pe250_c2": (dict(case="C2_adatrim_lowq", n=700, L=250), {}, "ILi8ELb0ELb1ELi16ENS_9TileShapeILi256"),
    "pe200_c3_runtime_shape": (dict(case="C3_full", n=700, L=200, var_len=True), {}, "ILi8ELb1ELb1ELi16ENS_9TileShapeILi0"),
    "pe150_runtime_shape": (dict(case="C3_full", n=700, L=150, var_len=True), {"SNK_TILED_RUNTIME_SHAPE": "1"}, "ILi5ELb1ELb1ELi16ENS_9TileShapeILi0"),
    "se100_c3": (dict(case="C3_full", n=1100, L=100, paired=False, var_len=True), {}, "ILi4ELb1ELb1"),
    "pe50_c2": (dict(case="C2_adatrim_lowq", n=700, L=50), {}, "ILi2ELb0ELb1"),
    "pe180_c3": (dict(case="C3_full", n=700, L=180, var_len=True), {}, "ILi6ELb1ELb1"),
    "pitch152_register_path": (dict(case="C3_full", n=701, L=150, pitch=152, var_len=True, first=1), {}, "ILi5ELb1ELb0"),
    "pitch152_c2_register_path": (dict(case="C2_adatrim_lowq", n=700, L=150, pitch=152), {}, "ILi5ELb0ELb0"),
    "lower_case_and_many_n": (dict(case="C3_full", n=700, L=150, var_len=True, lower=0.06), {}, "ILi5ELb1ELb1"),
    "meanq_polyx": (dict(case="meanq_polyx", n=700, L=150), {}, "ILi5ELb1ELb1"),
    "hard_lq_trim": (dict(case="hard_lq_trim", n=700, L=150, var_len=True), {}, "ILi5E"),
    "adapter_lists": (dict(case="multi_adapter_params2", n=700, L=150, dimer_frac=0.3), {}, "ILi5E"),
    "short_adapter_edge": (dict(case="short_adapter_edge", n=700, L=150, dimer_frac=0.3), {}, "ILi5E"),
    "many_flushes_one_workgroup": (dict(case="C3_full", n=3000, L=150, var_len=True), {"SNK_TEST_MAX_WGS": "1", "SNK_TEST_FLUSH_EVERY": "1"}, "ILi5ELb1ELb1"),
}
# an ordinary run (see tests/conftest.py: SNK_SIMT_FULL=1 takes everything)
CORE = ["test_assembly_matches_the_emulated_twin[pe150_c3_full_ragged]", "test_a_weakened_wait_is_caught", "test_a_scalar_load_of_a_dword_written",
        "test_the_bit_transposes_from_the_assembly"]


def simt_lib_path():
    import simt_lib as S
    return S.build_module().build()


def capture(tmp, spec, env=None, cus=2, kernels=("snk_tiled",)):
    lib = simt_lib_path()
    offs = ",".join("%x" % o for k in kernels for o in G.kernel_offsets(lib, k))
    e = dict(os.environ, SIMT_DUMP_DIR=str(tmp), SIMT_DUMP_OFFSETS=offs, SIMT_CUS=str(cus),
             PYTHONPATH=os.pathsep.join([HERE, T.ROOT, os.environ.get("PYTHONPATH", "")]))
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, "isa_interp_capture.py"), json.dumps(spec)], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "captured" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    return sorted(int(f[1:-5]) for f in os.listdir(tmp) if f.endswith(".json"))


needs_asm = pytest.mark.skipif(not os.path.exists(ASM), reason="the build's kept assembly is not there (python __graft_entry__.py)")


@needs_asm
@pytest.mark.parametrize("name", list(CASES))
def test_assembly_matches_the_emulated_twin(name, tmp_path):
    spec, env, instance = CASES[name]
    launches = capture(tmp_path, spec, env)
    assert len(launches) >= 2                     # the tiled kernel and the reduce kernel behind it
    seen = []
    for k in launches:
        info, diffs = G.replay(str(tmp_path), k, ASM, verbose=False, garbage=1)      # (registers start as noise, not zeros)
        seen.append(info["symbol"])
        assert info["instructions"] > 0
        assert not diffs, (info, diffs)
        assert info["scalar_loads_of_words_written_in_this_launch"] == 0, info      # (the scalar cache is not coherent inside a launch)
    assert any(instance in s for s in seen), seen
    assert any("snk_tiled_reduce_kernel" in s for s in seen), seen


# Round 5's envelopes that have met neither the hardware nor (until this test) their own instructions: the contexts of the GPU
# tier's first-contact tests, smaller.  name -> (capture spec, kernels captured, kernel names that must have been replayed)
BUILD = os.path.dirname(ASM)
ENVELOPES = {
    "adapters_shorter_than_6_pe250": (dict(any_length=0, n=500), ("snk_tiled",), ["snk_tiled_kernelILi8"]),
    "adapters_of_65_to_120": (dict(any_length=1, n=500), ("snk_tiled",), ["snk_tiled_kernelILi5"]),
    "adapter_of_130_to_200_and_a_short_one": (dict(any_length=2, n=500), ("snk_tiled",), ["snk_tiled_kernelILi5"]),
    "ada_edge_beyond_the_adapter_pe250": (dict(any_length=3, n=500), ("snk_tiled",), ["snk_tiled_kernelILi8"]),
    "long_reads_600_adapter_100": (dict(long_any_length=[600, 100, 6, 0.5], n=192, kernel=2), ("snk_long",),
                                   ["snk_long_prep_kernel", "snk_long_decide_kernel", "snk_long_hist_kernel"]),
    "long_reads_600_adapter_3": (dict(long_any_length=[600, 3, 2, 0.7], n=192, kernel=2), ("snk_long",), ["snk_long_decide_kernel"]),
    "long_reads_1000_adapter_255": (dict(long_any_length=[1000, 255, 10, 0.3], n=128, kernel=2), ("snk_long",), ["snk_long_decide_kernel"]),
    # the generic kernel alone (kernel = 1) on reads of 300 positions with contaminant lists: the sequential matchers in its lanes
    "contaminants_300_positions_generic_kernel": (dict(case="C2_adatrim_lowq", n=200, L=300, contam="single", kernel=1, var_len=True), ("snk_contam", "snk_generic"),
                                                  ["snk_generic_kernel"]),
    "long_reads_600_ada_edge_60_of_40": (dict(long_any_length=[600, 40, 60, 0.5], n=192, kernel=2), ("snk_long",), ["snk_long_decide_kernel"]),
}


def _replay(args):
    d, k = args
    info, diffs = G.replay(d, k, BUILD, verbose=False, garbage=2)
    if info["scalar_loads_of_words_written_in_this_launch"]:
        diffs = diffs + [("scalar loads of words written in this launch", info["scalar_loads_of_words_written_in_this_launch"])]
    return k, info["symbol"], info["instructions"], diffs


@needs_asm
@pytest.mark.parametrize("name", list(ENVELOPES))
def test_first_contact_envelopes_from_the_assembly(name, tmp_path):
    import concurrent.futures
    spec, kernels, expect = ENVELOPES[name]
    launches = capture(tmp_path, spec, kernels=kernels)
    assert launches
    with concurrent.futures.ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(_replay, [(str(tmp_path), k) for k in launches]))
    bad = [(k, sym, diffs) for k, sym, n, diffs in results if diffs or n == 0]
    assert not bad, bad
    for want in expect:
        assert any(want in sym for _, sym, _, _ in results), (want, [r[1] for r in results])


def mutated(tmp_path, symbol, edit):
    """a copy of the assembly with `edit` applied to the text of one function"""
    text = open(ASM).read()
    a = text.find("\n" + symbol + ":")
    b = text.find("s_endpgm", a)
    body = edit(text[a:b])
    out = os.path.join(str(tmp_path), "mutated.s")
    open(out, "w").write(text[:a] + body + text[b:])
    return out


@pytest.fixture
def c2_capture(tmp_path):
    capture(tmp_path, CASES["pe150_c2_static"][0])
    meta = json.load(open(os.path.join(str(tmp_path), "L0.json")))
    return str(tmp_path), G.symbol_at(meta["lib"], meta["offset"])


@needs_asm
def test_a_weakened_wait_is_caught(tmp_path, c2_capture):
    """the counted waits of phase 1's loop, each one entry too generous: a read's result is used before the wait that covers it"""
    d, sym = c2_capture

    def edit(body):
        body, n = re.subn(r"s_waitcnt lgkmcnt\(([1-9]\d*)\)", lambda m: "s_waitcnt lgkmcnt(%d)" % (int(m.group(1)) + 1), body)
        assert n >= 8
        return body

    with pytest.raises(G.Hazard, match="before the wait"):
        G.replay(d, 0, mutated(tmp_path, sym, edit), verbose=False)


@needs_asm
def test_a_missing_dma_wait_is_caught(tmp_path, c2_capture):
    """no vmcnt wait between the LDS DMA of a chunk and the LDS reads of its rows"""
    d, sym = c2_capture

    def edit(body):
        body, n = re.subn(r"s_waitcnt vmcnt\(\d+\)\n", "s_nop 0\n", body)
        assert n >= 8
        return body

    # (a vector load's register used early is a hazard; rows read from LDS under their DMA come back as garbage: a different image)
    try:
        info, diffs = G.replay(d, 0, mutated(tmp_path, sym, edit), verbose=False)
    except G.Hazard:
        return
    assert diffs and info["lds_bytes_read_under_a_dma"] > 0


@needs_asm
def test_a_changed_instruction_is_caught(tmp_path, c2_capture):
    """the two data operands of the v_bfi_b32 of the bit transposes exchanged: the memory image differs from the emulated twin's"""
    d, sym = c2_capture

    def edit(body):
        body, n = re.subn(r"v_bfi_b32 (v\d+), (v\d+), (v\d+), (v\d+)", r"v_bfi_b32 \1, \2, \4, \3", body)
        assert n >= 8
        return body

    info, diffs = G.replay(d, 0, mutated(tmp_path, sym, edit), verbose=False)
    assert diffs


@needs_asm
def test_a_missing_initialisation_is_caught_only_with_noise_in_the_registers(tmp_path, c2_capture):
    """the kernel's first `v_mov_b32 v, 0` (the value its LDS words are cleared with) taken out: with registers that start as zeros
    the replay still leaves the right memory -- which is why the replays start them as noise (--garbage)"""
    d, sym = c2_capture

    def edit(body):
        body, n = re.subn(r"\tv_mov_b32_e32 (v\d+), 0\n", "\ts_nop 0\n", body, count=1)
        assert n == 1
        return body

    asm = mutated(tmp_path, sym, edit)
    info, diffs = G.replay(d, 0, asm, verbose=False)
    assert not diffs                                  # (zeros happen to be what the code wanted)
    info, diffs = G.replay(d, 0, asm, verbose=False, garbage=1)
    assert diffs


@needs_asm
def test_the_bit_transposes_from_the_assembly(tmp_path):
    """bittr_selftest_kernel (64 x 64 bit transposes: DPP row shifts, v_permlane32_swap / v_permlane16_swap, v_bfi -- the hand-over of the
    tiled kernel uses the same functions) on random matrices: the instructions leave the words the emulated twin left, which the
    capture checks against a numpy transpose"""
    lib = simt_lib_path()
    code = """
import ctypes as C, numpy as np, sys
import simt_lib as S
lib = S.lib()
rng = np.random.default_rng(4)
n = 6
m = (rng.random((n, 64, 64)) < 0.5).astype(np.uint64)
w = (np.uint64(1) << np.arange(32, dtype=np.uint64))
words = np.ascontiguousarray(np.stack([(m[..., :32] * w).sum(-1), (m[..., 32:] * w).sum(-1)], axis=-1).astype(np.uint32))
out = np.zeros_like(words); lo = np.zeros((n, 64), dtype=np.uint32)
rc = lib.snk_selftest_bit_transpose(0, words.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), lo.ctypes.data_as(C.c_void_p))
assert rc == 0
got = ((out[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8).reshape(n, 64, 64)
assert np.array_equal(got, m.transpose(0, 2, 1).astype(np.uint8))
print("captured")
"""
    offs = ",".join("%x" % o for o in G.kernel_offsets(lib, "bittr_selftest"))
    e = dict(os.environ, SIMT_DUMP_DIR=str(tmp_path), SIMT_DUMP_OFFSETS=offs, PYTHONPATH=os.pathsep.join([HERE, T.ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "captured" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    for sch in (None, "reverse"):
        info, diffs = G.replay(str(tmp_path), 0, BUILD, verbose=False, garbage=4, schedule=sch)
        assert "bittr_selftest_kernel" in info["symbol"] and info["instructions"] > 0 and not diffs, (info, diffs)


def test_a_scalar_load_of_a_dword_written_in_the_launch_is_reported():
    """the mechanism behind `scalar_loads_of_words_written_in_this_launch` (no kernel of the build does it, so no capture shows it):
    a vector store, then scalar loads of the neighbouring dword (fine: same 64-byte line, other dword) and of the stored one (reported)"""
    import numpy as np
    mem = G.Memory()
    mem.add(0x1000, bytes(256))
    lanes = np.zeros(64, dtype=bool)
    lanes[3] = True
    addrs = np.zeros(64, dtype=np.uint64)
    addrs[3] = 0x1000 + 40
    mem.write(addrs, np.full((64, 4), 0xAB, dtype=np.uint8), lanes)
    mem.read_scalar(0x1000 + 44, 4)
    assert not mem.stale_scalar_reads and (0x1000 >> 6) in (mem.scalar_lines & mem.written_lines)
    mem.read_scalar(0x1000 + 32, 16)
    assert mem.stale_scalar_reads == [(0x1000 + 32, 16)]
