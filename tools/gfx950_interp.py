#!/usr/bin/env python3
"""Runs the gfx950 ASSEMBLY of a kernel -- the text hipcc keeps next to the object the product ships (-save-temps) -- on the CPU,
one wavefront of 64 lanes at a time, on a memory image captured from a launch of the SIMT emulator (tests/simt, SIMT_DUMP_DIR), and
compares what the instructions leave in memory with what the emulated C++ twin left there.

What this adds to the emulated tier (which compiles the HIP sources for the host): the instructions themselves.  The hand-placed
blocks (ds_read / ds_add with immediate offsets, counted s_waitcnt, the LDS DMA, DPP / permlane transposes, v_writelane spills,
the scalar flag) and everything the compiler made of the rest are executed as the ISA defines them, not as a C++ stand-in says.

Memory-ordering model (the part a counted wait can get wrong): a result that comes back asynchronously -- LDS reads, scalar loads,
vector loads, LDS DMA -- is written at once but its destination registers (for the DMA: its LDS bytes) stay POISONED until an
s_waitcnt covers them by the counters' rules: vmcnt and LDS-only lgkmcnt retire in order (the oldest len - N entries are done at
a wait for N), scalar loads return out of order (with one outstanding only a wait for 0 retires anything).  Reading or overwriting a
poisoned register, or an LDS read of poisoned bytes, is reported as a hazard with its line in the assembly.

No GPU, no ROCm runtime: numpy only.  Coverage is the instruction set the tiled kernel's instances use; an unknown mnemonic stops
the run with its name.
"""
import bisect
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np

U32 = np.uint32
U64 = np.uint64
I32 = np.int32
I64 = np.int64
LANES = np.arange(64, dtype=np.int64)
M32 = 0xFFFFFFFF

VCC, M0, EXEC = 106, 124, 126          # SGPR file indices of the special registers (as the ISA numbers them)


class Hazard(Exception):
    pass


class Unknown(Exception):
    pass


# ------------------------------------------------------------------------------------------------------------ parsing
MOD_KV = re.compile(r"\b([a-z_][a-z_0-9]*):(\[[^\]]*\]|\S+)")
MOD_FLAG = re.compile(r"\b(sc0|sc1|nt|glc|slc|clamp|lds|row_mirror|row_half_mirror)\b")
REG1 = re.compile(r"^([vsa])(\d+)$")
REGN = re.compile(r"^([vsa])\[(\d+):(\d+)\]$")
SPECIAL = {"vcc": (VCC, 2), "vcc_lo": (VCC, 1), "vcc_hi": (VCC + 1, 1), "exec": (EXEC, 2), "exec_lo": (EXEC, 1), "exec_hi": (EXEC + 1, 1),
           "m0": (M0, 1)}
FLOATS = {"0.5": 0x3F000000, "-0.5": 0xBF000000, "1.0": 0x3F800000, "-1.0": 0xBF800000, "2.0": 0x40000000, "-2.0": 0xC0000000,
          "4.0": 0x40800000, "-4.0": 0xC0800000}


class Op:
    __slots__ = ("kind", "n", "cnt", "val", "text")

    def __init__(self, kind, n=0, cnt=1, val=0, text=""):
        self.kind, self.n, self.cnt, self.val, self.text = kind, n, cnt, val, text

    def __repr__(self):
        return self.text


def parse_operand(t):
    t = t.strip()
    m = REG1.match(t)
    if m:
        return Op(m.group(1), int(m.group(2)), 1, text=t)
    m = REGN.match(t)
    if m:
        return Op(m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1, text=t)
    if t in SPECIAL:
        n, c = SPECIAL[t]
        return Op("s", n, c, text=t)
    if t == "off":
        return Op("off", text=t)
    if t in FLOATS:
        return Op("c", val=FLOATS[t], text=t)
    if t == "src_scc" or t == "scc":
        return Op("scc", text=t)
    try:
        return Op("c", val=int(t, 0) & 0xFFFFFFFFFFFFFFFF, text=t)
    except ValueError:
        pass
    m = re.match(r"^\((\S+)-(\S+)\)(&4294967295|>>32)$", t)
    if m:                                   # a long branch: the halves of the distance between two labels
        return Op("diff", n=1 if m.group(3) == ">>32" else 0, val=m.group(2), text=m.group(1))
    m = re.match(r"^(\S+)@rel32@(lo|hi)\+\d+$", t)
    if m:                                   # the two halves of `symbol - (the pc s_getpc_b64 returned)`, see ex_call
        return Op("rel", n=1 if m.group(2) == "hi" else 0, text=m.group(1))
    m = re.match(r"^(\S+)@gotpcrel32@(lo|hi)\+\d+$", t)
    if m:                                   # the same for the symbol's entry in the global offset table (an s_load_dwordx2 fetches the address)
        return Op("got", n=1 if m.group(2) == "hi" else 0, text=m.group(1))
    return Op("label", text=t)


class Ins:
    __slots__ = ("mn", "base", "ops", "mods", "flags", "line", "text", "fn")


SUFFIX = re.compile(r"_(e32|e64|dpp|sdwa)$")


def split_commas(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


_FILES = {}
CODE_BASE = 0x7D0000000000          # instruction k of a file "sits" at CODE_BASE + 4 k (s_getpc_b64 / s_swappc_b64 / s_setpc_b64)


def parse_file(asm_path):
    """-> (list of Ins of the whole file, labels {name: index} -- function symbols included --, text); cached"""
    key = (asm_path, os.path.getmtime(asm_path))
    if key in _FILES:
        return _FILES[key]
    text = open(asm_path).read()
    code = text[:text.find(".amdgpu_metadata")] if ".amdgpu_metadata" in text else text
    prog, labels = [], {}
    for k, raw in enumerate(code.split("\n")):
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            if line.endswith(":"):
                labels[line[:-1]] = len(prog)
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(prog)
            continue
        prog.append(make_ins(line, k + 1))
    _FILES[key] = (prog, labels, text)
    return _FILES[key]


def make_ins(line, lineno):
    """one instruction from its text (comments stripped)"""
    parts = line.split(None, 1)
    ins = Ins()
    ins.mn = parts[0]
    ins.base = SUFFIX.sub("", ins.mn)
    ins.line = lineno
    ins.text = line
    rest = parts[1] if len(parts) > 1 else ""
    ins.mods = {}
    ins.flags = set()
    if ins.base == "s_waitcnt":
        for m in re.finditer(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", rest):
            ins.mods[m.group(1)] = int(m.group(2))
        ins.ops = []
    else:
        for m in MOD_KV.finditer(rest):
            ins.mods[m.group(1)] = m.group(2)
        rest = MOD_KV.sub("", rest)
        m = re.search(r"gpr_idx\((\w+)\)", rest)
        if m:
            ins.mods["gpr_idx"] = m.group(1)
            rest = rest.replace(m.group(0), "")
        for m in MOD_FLAG.finditer(rest):
            ins.flags.add(m.group(1))
        rest = MOD_FLAG.sub("", rest)
        ins.ops = [parse_operand(t) for t in split_commas(rest) if t.strip()]
    try:
        ins.fn = handler(ins)
    except Unknown:
        ins.fn = ex_unknown
    return ins


DATA_BASE = 0x7C0000000000          # the file's data sections (constant tables the compiler emits, device variables with initialisers)
GOT_BASE = 0x7B0000000000           # one 8-byte entry per symbol taken through @gotpcrel32
_DATA = {}
_INT_DIRECTIVES = {".byte": 1, ".short": 2, ".2byte": 2, ".hword": 2, ".long": 4, ".4byte": 4, ".int": 4, ".quad": 8, ".8byte": 8}


def _c_string(lit):
    out, i = bytearray(), 0
    esc = {"n": 10, "t": 9, "r": 13, "b": 8, "f": 12, "\\": 92, '"': 34, "v": 11, "a": 7}
    while i < len(lit):
        ch = lit[i]
        if ch != "\\":
            out.append(ord(ch))
            i += 1
            continue
        i += 1
        m = re.match(r"[0-7]{1,3}", lit[i:])
        if m:
            out.append(int(m.group(0), 8) & 0xFF)
            i += len(m.group(0))
        elif lit[i] == "x":
            m = re.match(r"[0-9a-fA-F]+", lit[i + 1:])
            out.append(int(m.group(0), 16) & 0xFF)
            i += 1 + len(m.group(0))
        else:
            out.append(esc.get(lit[i], ord(lit[i])))
            i += 1
    return bytes(out)


def parse_data(asm_path):
    """the data sections of the file: -> ({symbol: offset into the image}, image bytes).  Code sections are skipped; a value that is
    not a number (a relocation against another symbol) is stored as zero."""
    key = (asm_path, os.path.getmtime(asm_path))
    if key in _DATA:
        return _DATA[key]
    text = open(asm_path).read()
    code = text[:text.find(".amdgpu_metadata")] if ".amdgpu_metadata" in text else text
    syms, img, in_data = {}, bytearray(), False
    for raw in code.split("\n"):
        line = raw.strip()
        if not line:
            continue
        if line.startswith(".text"):
            in_data = False
            continue
        if line.startswith((".data", ".bss", ".rodata")):
            in_data = True
            continue
        if line.startswith(".section"):
            name = line.split()[1].split(",")[0].strip('"')
            in_data = name.startswith((".rodata", ".data", ".bss")) and '"ax"' not in line and "amdhsa" not in name
            continue
        if not in_data:
            continue
        if line.startswith('.ascii') or line.startswith('.asciz') or line.startswith('.string'):
            d, rest = line.split(None, 1)
            for m in re.finditer(r'"((?:[^"\\]|\\.)*)"', rest):
                img += _c_string(m.group(1)) + (b"\0" if d != ".ascii" else b"")
            continue
        line = line.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            syms[line[:-1]] = len(img)
            continue
        if line.endswith(":"):
            syms[line[:-1]] = len(img)
            continue
        p = line.split(None, 1)
        d, rest = p[0], (p[1] if len(p) > 1 else "")
        if d in _INT_DIRECTIVES:
            for t in rest.split(","):
                try:
                    v = int(t.strip(), 0)
                except ValueError:
                    v = 0
                img += (v & ((1 << (8 * _INT_DIRECTIVES[d])) - 1)).to_bytes(_INT_DIRECTIVES[d], "little")
        elif d in (".zero", ".space", ".skip"):
            a = [int(t.strip(), 0) for t in rest.split(",")]
            img += bytes([a[1] & 0xFF if len(a) > 1 else 0]) * a[0]
        elif d == ".fill":
            a = [int(t.strip(), 0) for t in rest.split(",")] + [1, 0]
            img += (a[2] & ((1 << (8 * a[1])) - 1)).to_bytes(a[1], "little") * a[0]
        elif d in (".p2align", ".align", ".balign"):
            a = int(rest.split(",")[0].strip(), 0)
            a = 1 << a if d == ".p2align" else max(a, 1)
            img += bytes(-len(img) % a)
    _DATA[key] = (syms, bytes(img))
    return _DATA[key]


def parse_function(asm_path, symbol):
    """-> (the file's program, labels, the kernel descriptor of `symbol` {key: int})"""
    prog, labels, text = parse_file(asm_path)
    if symbol not in labels:
        raise KeyError("no function %s in %s" % (symbol, asm_path))
    kd = {}
    m = re.search(r"\.amdhsa_kernel " + re.escape(symbol) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
    if m:
        for l in m.group(1).split("\n"):
            p = l.split()
            if len(p) == 2 and p[0].startswith(".amdhsa_"):
                try:
                    kd[p[0][8:]] = int(p[1], 0)
                except ValueError:
                    pass
    return prog, labels, kd


def find_asm(build_dir, symbol):
    """the device assembly file (of the build's kept temporaries) that defines `symbol`"""
    import glob
    for f in sorted(glob.glob(os.path.join(build_dir, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        if ("\n" + symbol + ":") in open(f).read():
            return f
    raise KeyError("no assembly for " + symbol)


def kernarg_segment(asm_path, symbol, explicit, grid, block, shmem):
    """the explicit arguments + the hidden ones the code object's metadata lists for the kernel (block counts, group sizes, ...)"""
    text = open(asm_path).read()
    md = text[text.find(".amdgpu_metadata"):]
    entry = None
    for e in re.split(r"\n  - (?=\.agpr_count)", md):
        if re.search(r"\.name:\s+" + re.escape(symbol) + r"\n", e):
            entry = e
    if entry is None:
        raise KeyError("no metadata for " + symbol)
    size = int(re.search(r"\.kernarg_segment_size:\s+(\d+)", entry).group(1))
    seg = bytearray(max(size, len(explicit)))
    seg[:len(explicit)] = explicit
    for m in re.finditer(r"\.offset:\s+(\d+)\n\s+\.size:\s+(\d+)\n\s+\.value_kind:\s+(\w+)", entry):
        off, n, kind = int(m.group(1)), int(m.group(2)), m.group(3)
        if not kind.startswith("hidden_"):
            continue
        k = kind[7:]
        val = 0
        for ax, name in enumerate("xyz"):
            if k == "block_count_" + name:
                val = grid[ax]
            if k == "group_size_" + name:
                val = block[ax]
        if k == "grid_dims":
            val = 1 + (grid[1] * block[1] > 1) + (grid[2] * block[2] > 1)
        if k == "dynamic_lds_size":
            val = shmem
        seg[off:off + n] = int(val).to_bytes(n, "little")
    return bytes(seg)


# ------------------------------------------------------------------------------------------------------------ memory
class Memory:
    """the captured allocations at their captured addresses"""

    def __init__(self):
        self.bases, self.sizes, self.bufs = [], [], []
        self.got = {}                       # symbol -> its entry's address in the table at GOT_BASE
        self.symbol_address = None          # name -> address of a device variable (set by the caller; replay(): the emulated library's own copy)
        # the scalar cache is not coherent with the vector path inside a launch: 64-byte lines a scalar load read, lines a vector
        # store / atomic wrote.  A line in both sets is memory the kernel reads as constant and also changes (run_launch reports them)
        self.scalar_lines, self.written_lines = set(), set()
        self.written_words = set()          # dwords a vector store / atomic of this launch wrote
        self.stale_scalar_reads = []        # (address, bytes) of scalar loads that read such a dword: the scalar cache may hold the old line

    def got_entry(self, name):
        if name not in self.got:
            if GOT_BASE not in self.bases:
                self.add(GOT_BASE, bytes(8 * 512))
            addr = self.symbol_address(name) if self.symbol_address else None
            if addr is None:
                raise Unknown("no address for the device variable " + name)
            at = GOT_BASE + 8 * len(self.got)
            j = self.bases.index(GOT_BASE)
            self.bufs[j][at - GOT_BASE:at - GOT_BASE + 8] = np.frombuffer(int(addr).to_bytes(8, "little"), dtype=np.uint8)
            self.got[name] = at
        return self.got[name]

    def add(self, base, data):
        i = bisect.bisect_left(self.bases, base)
        self.bases.insert(i, base)
        self.sizes.insert(i, len(data))
        self.bufs.insert(i, np.frombuffer(bytearray(data), dtype=np.uint8))

    def find(self, lo, hi, what):
        i = bisect.bisect_right(self.bases, lo) - 1
        if i < 0 or hi > self.bases[i] + self.sizes[i]:
            raise Hazard("%s of [%#x, %#x): outside every captured allocation" % (what, lo, hi))
        return i

    def read(self, addrs, nbytes, mask, what="load"):
        """addrs uint64[64] -> uint8[64, nbytes] (zeros for inactive lanes)"""
        out = np.zeros((64, nbytes), dtype=np.uint8)
        if not mask.any():
            return out
        a = addrs[mask].astype(np.int64)
        lo, hi = int(a.min()), int(a.max()) + nbytes
        i = bisect.bisect_right(self.bases, lo) - 1
        if i >= 0 and hi <= self.bases[i] + self.sizes[i]:
            idx = (a - self.bases[i])[:, None] + np.arange(nbytes)[None, :]
            out[mask] = self.bufs[i][idx]
            return out
        for l in np.nonzero(mask)[0]:
            x = int(addrs[l])
            j = self.find(x, x + nbytes, what)
            out[l] = self.bufs[j][x - self.bases[j]:x - self.bases[j] + nbytes]
        return out

    def write(self, addrs, data, mask, what="store"):
        nbytes = data.shape[1]
        for l in np.nonzero(mask)[0]:
            x = int(addrs[l])
            j = self.find(x, x + nbytes, what)
            self.bufs[j][x - self.bases[j]:x - self.bases[j] + nbytes] = data[l]
            self.written_lines.update(range(x >> 6, ((x + nbytes - 1) >> 6) + 1))
            self.written_words.update(range(x >> 2, ((x + nbytes - 1) >> 2) + 1))

    def read_scalar(self, addr, nbytes):
        j = self.find(addr, addr + nbytes, "scalar load")
        self.scalar_lines.update(range(addr >> 6, ((addr + nbytes - 1) >> 6) + 1))
        if self.written_words and not self.written_words.isdisjoint(range(addr >> 2, ((addr + nbytes - 1) >> 2) + 1)):
            self.stale_scalar_reads.append((addr, nbytes))
        return bytes(self.bufs[j][addr - self.bases[j]:addr - self.bases[j] + nbytes])


# ------------------------------------------------------------------------------------------------------------ the wave
def u32(x):
    return np.asarray(x).astype(U32)


def s32(a):
    return a.view(I32)


class Wave:
    def __init__(self, wg, wave_id, prog, labels, mem, kd):
        self.wg, self.id, self.prog, self.labels, self.mem = wg, wave_id, prog, labels, mem
        self.s = np.zeros(128, dtype=np.uint64)          # 32-bit values (kept in 64-bit cells: plain python ints come out of it without overflow warnings)
        self.v = np.zeros((512, 64), dtype=U32)
        self.scc = 0
        self.pc = 0
        self.getpc = 0
        self.done = False
        self.idx_dst = None                              # s_set_gpr_idx_on ... gpr_idx(DST): offset added to a VALU destination
        self.idx_src0 = None                             # ... gpr_idx(SRC0): to a VALU instruction's first source
        self.lgkm = []                                   # outstanding: ("lds" | "smem", [poisoned registers])
        self.vm = []                                     # outstanding: ("load" | "store" | "dma", registers / LDS range)
        self.pv = {}                                     # poisoned VGPR -> line of the load
        self.ps = {}                                     # poisoned SGPR -> line
        self.nexec = 0
        self.count = {}
        self._exec_cache = (None, None)
        self.cur = None
        self.scratch = np.full((64, kd.get("private_segment_fixed_size", 0) + 16), 0xEE, dtype=np.uint8)
        self.data_syms = {}                              # data symbols of the file (parse_data): offsets behind DATA_BASE
        self.cov = None                                  # coverage: one counter per instruction of the file (run_launch(coverage=...))
        self.cap = None                                  # instruction budget of this wave (mutants that never leave a loop: run_launch(max_wave_instructions=...))

    # ---- registers
    def sget(self, n):
        return int(self.s[n])

    def sset(self, n, v):
        self.s[n] = v & M32

    def sget64(self, n):
        return int(self.s[n]) | (int(self.s[n + 1]) << 32)

    def sset64(self, n, v):
        self.s[n] = v & M32
        self.s[n + 1] = (v >> 32) & M32

    @property
    def exec(self):
        return self.sget64(EXEC)

    def execm(self):
        e = self.exec
        if self._exec_cache[0] != e:
            self._exec_cache = (e, np.array([(e >> l) & 1 for l in range(64)], dtype=bool))
        return self._exec_cache[1]

    def chk_v(self, n, cnt=1):
        if self.pv:
            for k in range(n, n + cnt):
                if k in self.pv:
                    raise Hazard("v%d is used at line %d (%s) before the wait that covers its load at line %d" % (k, self.cur.line, self.cur.text, self.pv[k]))

    def chk_s(self, n, cnt=1):
        if self.ps:
            for k in range(n, n + cnt):
                if k in self.ps:
                    raise Hazard("s%d is used at line %d (%s) before the wait that covers its load at line %d" % (k, self.cur.line, self.cur.text, self.ps[k]))

    def rd_s(self, op):
        """scalar operand (SGPR, constant) as a python int, 32 bits"""
        if op.kind == "s":
            self.chk_s(op.n)
            return int(self.s[op.n])
        if op.kind == "c":
            return op.val & M32
        if op.kind == "scc":
            return self.scc
        if op.kind == "diff":
            d = (4 * (self.labels[op.text] - self.labels[op.val])) & 0xFFFFFFFFFFFFFFFF
            return (d >> 32) & M32 if op.n else d & M32
        if op.kind == "rel":
            at = DATA_BASE + self.data_syms[op.text] if op.text in self.data_syms else CODE_BASE + 4 * self.labels[op.text]
            d = (at - self.getpc) & 0xFFFFFFFFFFFFFFFF
            return (d >> 32) & M32 if op.n else d & M32
        if op.kind == "got":
            d = (self.mem.got_entry(op.text) - self.getpc) & 0xFFFFFFFFFFFFFFFF
            return (d >> 32) & M32 if op.n else d & M32
        raise Unknown("scalar operand %r in %s" % (op, self.cur.text))

    def rd_s64(self, op):
        if op.kind == "s":
            self.chk_s(op.n, 2)
            return self.sget64(op.n)
        if op.kind == "c":
            v = op.val
            if op.text.startswith("-"):
                v = int(op.text, 0) & 0xFFFFFFFFFFFFFFFF
            return v & 0xFFFFFFFFFFFFFFFF
        raise Unknown("scalar operand %r in %s" % (op, self.cur.text))

    def rd32(self, op):
        """vector source as uint32[64]"""
        if op.kind == "v":
            self.chk_v(op.n)
            return self.v[op.n].copy()          # (a copy: the destination may be the same register)
        return np.full(64, self.rd_s(op), dtype=U32)

    def rd64(self, op):
        if op.kind == "v":
            self.chk_v(op.n, 2)
            return self.v[op.n].astype(U64) | (self.v[op.n + 1].astype(U64) << U64(32))
        return np.full(64, self.rd_s64(op), dtype=U64)

    def wr32(self, op, val, mask=None):
        n = op.n
        if self.idx_dst is not None:
            n += self.idx_dst
        if n in self.pv:
            raise Hazard("v%d is overwritten at line %d (%s) while its load at line %d is outstanding" % (n, self.cur.line, self.cur.text, self.pv[n]))
        m = self.execm() if mask is None else mask
        self.v[n] = np.where(m, u32(val), self.v[n])

    def wr64(self, op, val, mask=None):
        val = np.asarray(val).astype(U64)
        m = self.execm() if mask is None else mask
        for k in range(2):
            if op.n + k in self.pv:
                raise Hazard("v%d overwritten while outstanding (line %d)" % (op.n + k, self.cur.line))
        self.v[op.n] = np.where(m, (val & U64(M32)).astype(U32), self.v[op.n])
        self.v[op.n + 1] = np.where(m, (val >> U64(32)).astype(U32), self.v[op.n + 1])

    def wr_mask(self, op, bits):
        """a lane mask (bool[64]) into an SGPR pair; inactive lanes give 0"""
        b = bits & self.execm()
        v = 0
        for l in np.nonzero(b)[0]:
            v |= 1 << int(l)
        self.sset64(op.n, v)

    def rd_mask(self, op):
        v = self.rd_s64(op)
        return np.array([(v >> l) & 1 for l in range(64)], dtype=bool)

    # ---- waits
    def retire(self, entry):
        kind, regs = entry
        if kind == "dma":
            self.wg.lds_poison[regs] -= 1
            return
        for (f, r) in regs:
            (self.pv if f == "v" else self.ps).pop(r, None)

    def waitcnt(self, ins):
        if "vmcnt" in ins.mods:
            n = ins.mods["vmcnt"]
            while len(self.vm) > n:
                self.retire(self.vm.pop(0))
        if "lgkmcnt" in ins.mods:
            n = ins.mods["lgkmcnt"]
            if n == 0:
                while self.lgkm:
                    self.retire(self.lgkm.pop(0))
            elif not any(e[0] == "smem" for e in self.lgkm):
                while len(self.lgkm) > n:
                    self.retire(self.lgkm.pop(0))
            # scalar loads outstanding: they return in any order, a wait for N > 0 guarantees none of the results

    def poison(self, queue, kind, regs):
        for (f, r) in regs:
            (self.pv if f == "v" else self.ps)[r] = self.cur.line
        queue.append((kind, regs))

    # ---- run
    def run(self, limit=None):
        """until the wave ends ('end') or reaches a barrier ('barrier')"""
        prog = self.prog
        cov = self.cov
        while True:
            ins = prog[self.pc]
            self.cur = ins
            if cov is not None:
                cov[self.pc] += 1
            self.pc += 1
            self.nexec += 1
            r = ins.fn(self, ins)
            if r is not None:
                return r
            if limit is not None and self.nexec >= limit:
                return "limit"
            if self.cap is not None and self.nexec > self.cap:
                raise Hazard("line %d: the wave is past its budget of %d instructions (a loop that does not end)" % (ins.line, self.cap))


# ------------------------------------------------------------------------------------------------------------ DPP / SDWA
def dpp_source(w, ins, src):
    """src0 through the DPP controls -> (value[64], lanes whose write is enabled)"""
    mods = ins.mods
    lane = LANES
    row = lane & ~15
    valid = np.ones(64, dtype=bool)
    if "quad_perm" in mods:
        p = [int(x) for x in mods["quad_perm"].strip("[]").split(",")]
        sl = (lane & ~3) + np.array(p, dtype=np.int64)[lane & 3]
    elif "row_shl" in mods:
        sl = lane + int(mods["row_shl"], 0)
        valid = (sl & ~15) == row
    elif "row_shr" in mods:
        sl = lane - int(mods["row_shr"], 0)
        valid = (sl >= 0) & ((sl & ~15) == row)
    elif "row_ror" in mods:
        sl = row + (((lane & 15) - int(mods["row_ror"], 0)) & 15)
    elif "row_mirror" in ins.flags:
        sl = row + 15 - (lane & 15)
    elif "row_half_mirror" in ins.flags:
        sl = (lane & ~7) + 7 - (lane & 7)
    elif "row_bcast" in mods:
        k = int(mods["row_bcast"], 0)
        if k == 15:
            sl = row - 1
            valid = lane >= 16
        else:
            sl = np.full(64, 31, dtype=np.int64)
            valid = lane >= 32
    elif "wave_shl" in mods:
        sl = lane + 1
        valid = sl < 64
    elif "wave_shr" in mods:
        sl = lane - 1
        valid = sl >= 0
    elif "wave_rol" in mods:
        sl = (lane + 1) & 63
    elif "wave_ror" in mods:
        sl = (lane - 1) & 63
    else:
        raise Unknown("DPP control in " + ins.text)
    sl = np.clip(sl, 0, 63)
    em = w.execm()
    valid = valid & em[sl]
    val = np.where(valid, src[sl], U32(0))
    rm, bm = int(mods.get("row_mask", "0xf"), 0), int(mods.get("bank_mask", "0xf"), 0)
    en = (((rm >> (lane >> 4)) & 1) == 1) & (((bm >> ((lane >> 2) & 3)) & 1) == 1)
    if int(mods.get("bound_ctrl", "0"), 0) == 0:
        en = en & valid
    return val, en & em


SEL = {"BYTE_0": (0, 0xFF), "BYTE_1": (8, 0xFF), "BYTE_2": (16, 0xFF), "BYTE_3": (24, 0xFF), "WORD_0": (0, 0xFFFF), "WORD_1": (16, 0xFFFF),
       "DWORD": (0, M32)}


def sdwa_src(ins, k, a):
    sel = ins.mods.get("src%d_sel" % k, "DWORD")
    sh, m = SEL[sel]
    return (a >> U32(sh)) & U32(m)


def sdwa_dst(ins, r):
    sel = ins.mods.get("dst_sel", "DWORD")
    if sel == "DWORD":
        return r
    if ins.mods.get("dst_unused", "UNUSED_PAD") != "UNUSED_PAD":
        raise Unknown("dst_unused in " + ins.text)
    sh, m = SEL[sel]
    return (r & U32(m)) << U32(sh)


# ------------------------------------------------------------------------------------------------------------ VALU
def popc32(a):
    a = a.astype(U32)
    a = a - ((a >> U32(1)) & U32(0x55555555))
    a = (a & U32(0x33333333)) + ((a >> U32(2)) & U32(0x33333333))
    a = (a + (a >> U32(4))) & U32(0x0F0F0F0F)
    return (a * U32(0x01010101)) >> U32(24)


def ffbl(a):
    r = np.full(a.shape, M32, dtype=U32)
    for b in range(31, -1, -1):
        r = np.where((a >> U32(b)) & U32(1) == 1, U32(b), r)
    return r


def ffbh(a):
    r = np.full(a.shape, M32, dtype=U32)
    for b in range(0, 32):
        r = np.where((a >> U32(b)) & U32(1) == 1, U32(31 - b), r)
    return r


def bfrev(a):
    r = np.zeros_like(a)
    for b in range(32):
        r |= ((a >> U32(b)) & U32(1)) << U32(31 - b)
    return r


def perm_b32(s0, s1, sel):
    """bytes of {s0, s1} (s1 = bytes 0..3, s0 = bytes 4..7) picked by the four selector bytes"""
    src = np.stack([(s1 >> U32(8 * k)) & U32(0xFF) for k in range(4)] + [(s0 >> U32(8 * k)) & U32(0xFF) for k in range(4)], axis=0)   # [8, 64]
    out = np.zeros(64, dtype=U32)
    for k in range(4):
        c = ((sel >> U32(8 * k)) & U32(0xFF)).astype(np.int64)
        b = np.where(c < 8, src[np.minimum(c, 7), LANES], 0).astype(U32)
        # 8..11: sign of bytes 1, 3, 5, 7 replicated; 12: 0x00; >= 13: 0xFF
        for q, by in ((8, 1), (9, 3), (10, 5), (11, 7)):
            b = np.where(c == q, np.where(src[by] & U32(0x80), U32(0xFF), U32(0)), b)
        b = np.where(c >= 13, U32(0xFF), b)
        out |= b.astype(U32) << U32(8 * k)
    return out


def bitop3(a, b, c, tt):
    r = np.zeros_like(a)
    for i in range(8):
        if (tt >> i) & 1:
            ta = a if i & 4 else ~a
            tb = b if i & 2 else ~b
            tc = c if i & 1 else ~c
            r |= ta & tb & tc
    return r


def med3(a, b, c):
    return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))


def sad_u8(a, b, c):
    r = c.copy()
    for k in range(4):
        x = ((a >> U32(8 * k)) & U32(0xFF)).astype(np.int64)
        y = ((b >> U32(8 * k)) & U32(0xFF)).astype(np.int64)
        r = r + np.abs(x - y).astype(U32)
    return r


def f32(a):
    return a.view(np.float32)


def cvt_u32_f32(a):
    f = f32(a).astype(np.float64)
    f = np.where(np.isnan(f), 0.0, f)
    return np.clip(np.trunc(f), 0, 4294967295.0).astype(np.uint64).astype(U32)


def sh(b):
    return b & U32(31)


# name -> function of the 32-bit sources (uint32[64] each) -> uint32[64]
VOP = {
    "v_mov_b32": lambda a: a,
    "v_not_b32": lambda a: ~a,
    "v_bfrev_b32": bfrev,
    "v_ffbl_b32": ffbl,
    "v_ffbh_u32": ffbh,
    "v_cvt_f32_u32": lambda a: a.astype(np.float32).view(U32),
    "v_cvt_u32_f32": cvt_u32_f32,
    "v_rcp_iflag_f32": lambda a: (np.float32(1.0) / f32(a)).astype(np.float32).view(U32),
    "v_mul_f32": lambda a, b: (f32(a) * f32(b)).astype(np.float32).view(U32),
    "v_rcp_f32": lambda a: (np.float32(1.0) / f32(a)).astype(np.float32).view(U32),
    "v_trunc_f32": lambda a: np.trunc(f32(a)).astype(np.float32).view(U32),
    "v_fmamk_f32": lambda a, k, c: (f32(a).astype(np.float64) * f32(k).astype(np.float64) + f32(c).astype(np.float64)).astype(np.float32).view(U32),
    "v_add_u32": lambda a, b: a + b,
    "v_sub_u32": lambda a, b: a - b,
    "v_subrev_u32": lambda a, b: b - a,
    "v_and_b32": lambda a, b: a & b,
    "v_or_b32": lambda a, b: a | b,
    "v_xor_b32": lambda a, b: a ^ b,
    "v_xnor_b32": lambda a, b: ~(a ^ b),
    "v_lshlrev_b32": lambda a, b: b << sh(a),
    "v_lshrrev_b32": lambda a, b: b >> sh(a),
    "v_ashrrev_i32": lambda a, b: (s32(b) >> sh(a).astype(I32)).view(U32),
    "v_min_i32": lambda a, b: np.minimum(s32(a), s32(b)).view(U32),
    "v_max_i32": lambda a, b: np.maximum(s32(a), s32(b)).view(U32),
    "v_min_u32": np.minimum,
    "v_max_u32": np.maximum,
    "v_mul_lo_u32": lambda a, b: a * b,
    "v_mul_hi_u32": lambda a, b: ((a.astype(U64) * b.astype(U64)) >> U64(32)).astype(U32),
    "v_mul_u32_u24": lambda a, b: (a & U32(0xFFFFFF)) * (b & U32(0xFFFFFF)),
    "v_bcnt_u32_b32": lambda a, b: popc32(a) + b,
    "v_alignbit_b32": lambda a, b, c: (((a.astype(U64) << U64(32)) | b.astype(U64)) >> (c & U32(31)).astype(U64)).astype(U32),
    "v_perm_b32": perm_b32,
    "v_bfe_u32": lambda a, b, c: np.where((c & U32(31)) == 0, U32(0), (a >> sh(b)) & ((U32(1) << sh(c)) - U32(1))),
    "v_bfi_b32": lambda a, b, c: (a & b) | (~a & c),
    "v_med3_u32": med3,
    "v_med3_i32": lambda a, b, c: med3(s32(a), s32(b), s32(c)).view(U32),
    "v_max3_i32": lambda a, b, c: np.maximum(np.maximum(s32(a), s32(b)), s32(c)).view(U32),
    "v_lshl_add_u32": lambda a, b, c: (a << sh(b)) + c,
    "v_add_lshl_u32": lambda a, b, c: (a + b) << sh(c),
    "v_lshl_or_b32": lambda a, b, c: (a << sh(b)) | c,
    "v_and_or_b32": lambda a, b, c: (a & b) | c,
    "v_or3_b32": lambda a, b, c: a | b | c,
    "v_add3_u32": lambda a, b, c: a + b + c,
    "v_xad_u32": lambda a, b, c: (a ^ b) + c,
    "v_mad_u32_u24": lambda a, b, c: (a & U32(0xFFFFFF)) * (b & U32(0xFFFFFF)) + c,
    "v_sad_u8": sad_u8,
    "v_sad_hi_u8": lambda a, b, c: (sad_u8(a, b, np.zeros(64, dtype=U32)) << U32(16)) + c,
    # 16-bit operations: the low halves, the high half of the destination cleared (gfx9)
    "v_add_u16": lambda a, b: (a + b) & U32(0xFFFF),
    "v_lshlrev_b16": lambda a, b: (b << (a & U32(15))) & U32(0xFFFF),
    "v_pk_add_u16": lambda a, b: ((a + b) & U32(0xFFFF)) | ((((a >> U32(16)) + (b >> U32(16))) & U32(0xFFFF)) << U32(16)),
}


def ex_valu(w, ins):
    fn = VOP[ins.base]
    ops = ins.ops
    srcs = [w.rd32(o) for o in ops[1:]]
    if w.idx_src0 is not None and ops[1].kind == "v":
        w.chk_v(ops[1].n + w.idx_src0)
        srcs[0] = w.v[ops[1].n + w.idx_src0].copy()
    mask = None
    if ins.mn.endswith("_dpp"):
        srcs[0], mask = dpp_source(w, ins, srcs[0])
    elif ins.mn.endswith("_sdwa"):
        srcs = [sdwa_src(ins, k, a) for k, a in enumerate(srcs)]
    r = fn(*srcs)
    if "clamp" in ins.flags:
        # the integer clamp of VOP3 / SDWA: the result saturates instead of wrapping (unsigned: 0 .. 2^32 - 1).  Found ignored in round 6
        # by the widened captures: `v_sub_u32_e64 v4, v82, v5 clamp` is how the compiler spells trim_finish()'s "cuts longer than the
        # read -> clean length 0", and no capture before had a read whose cuts exceeded it inside the tiled kernel.
        a, b = (x.astype(np.int64) for x in srcs[:2])
        if ins.base == "v_sub_u32":
            r = np.maximum(a - b, 0).astype(U32)
        elif ins.base == "v_subrev_u32":
            r = np.maximum(b - a, 0).astype(U32)
        elif ins.base == "v_add_u32":
            r = np.minimum(a + b, M32).astype(U32)
        else:
            raise Unknown("clamp on %s" % ins.text)
    if ins.mn.endswith("_sdwa"):
        r = sdwa_dst(ins, u32(r))
    w.wr32(ops[0], r, mask)


def ex_fmac(w, ins):
    a, b, c = w.rd32(ins.ops[1]), w.rd32(ins.ops[2]), w.rd32(ins.ops[0])
    w.wr32(ins.ops[0], (f32(a).astype(np.float64) * f32(b).astype(np.float64) + f32(c).astype(np.float64)).astype(np.float32).view(U32))


def ex_bitop3(w, ins):
    a, b, c = (w.rd32(o) for o in ins.ops[1:4])
    r = bitop3(a, b, c, int(ins.mods["bitop3"], 0))
    if ins.base.endswith("b16"):
        r = r & U32(0xFFFF)
    w.wr32(ins.ops[0], r)


CMP = {"eq": lambda a, b: a == b, "ne": lambda a, b: a != b, "lg": lambda a, b: a != b, "lt": lambda a, b: a < b, "le": lambda a, b: a <= b,
       "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b}


def ex_vcmp(w, ins):
    _, _, op, ty = ins.base.split("_")        # v_cmp_<op>_<type>
    wide = ty in ("u64", "i64")
    rd = w.rd64 if wide else w.rd32
    a, b = rd(ins.ops[1]), rd(ins.ops[2])
    if ins.mn.endswith("_sdwa"):
        a, b = sdwa_src(ins, 0, a), sdwa_src(ins, 1, b)
    if ty == "i32":
        a, b = s32(a), s32(b)
    elif ty == "i64":
        a, b = a.view(I64), b.view(I64)
    elif ty == "u16":
        a, b = a & U32(0xFFFF), b & U32(0xFFFF)
    elif ty == "i16":
        a, b = (a & U32(0xFFFF)).astype(np.uint16).view(np.int16), (b & U32(0xFFFF)).astype(np.uint16).view(np.int16)
    elif ty not in ("u32", "u64"):
        raise Unknown(ins.text)
    w.wr_mask(ins.ops[0], CMP[op](a, b))


def ex_cndmask(w, ins):
    a, b = w.rd32(ins.ops[1]), w.rd32(ins.ops[2])
    m = w.rd_mask(ins.ops[3])
    if ins.mn.endswith("_sdwa"):
        a, b = sdwa_src(ins, 0, a), sdwa_src(ins, 1, b)
        w.wr32(ins.ops[0], sdwa_dst(ins, np.where(m, b, a)))
        return
    if ins.mn.endswith("_dpp"):
        raise Unknown(ins.text)
    w.wr32(ins.ops[0], np.where(m, b, a))


def ex_carry(w, ins):
    """v_add_co / v_sub_co / v_subrev_co / v_subb_co (vdst, carry-out, a, b [, carry-in])"""
    a, b = w.rd32(ins.ops[2]).astype(np.int64), w.rd32(ins.ops[3]).astype(np.int64)
    cin = w.rd_mask(ins.ops[4]).astype(np.int64) if len(ins.ops) > 4 else 0
    base = ins.base
    if base.startswith("v_add"):
        r = a + b + cin
        c = r > M32
    elif base.startswith("v_subrev") or base.startswith("v_subbrev"):
        r = b - a - cin
        c = r < 0
    else:
        r = a - b - cin
        c = r < 0
    w.wr32(ins.ops[0], (r & M32).astype(U32))
    w.wr_mask(ins.ops[1], c)


def ex_mad64(w, ins):
    signed = ins.base == "v_mad_i64_i32"
    a, b = w.rd32(ins.ops[2]), w.rd32(ins.ops[3])
    c = w.rd64(ins.ops[4])
    if signed:
        p = (s32(a).astype(I64) * s32(b).astype(I64)).view(U64)
    else:
        p = a.astype(U64) * b.astype(U64)
    r = p + c
    w.wr64(ins.ops[0], r)
    w.wr_mask(ins.ops[1], r < p)          # (unsigned carry; nobody reads it in these kernels)


def ex_v64(w, ins):
    b = ins.base
    if b == "v_mov_b64":
        r = w.rd64(ins.ops[1])
    elif b == "v_lshlrev_b64":
        r = w.rd64(ins.ops[2]) << (w.rd32(ins.ops[1]) & U32(63)).astype(U64)
    elif b == "v_lshrrev_b64":
        r = w.rd64(ins.ops[2]) >> (w.rd32(ins.ops[1]) & U32(63)).astype(U64)
    elif b == "v_lshl_add_u64":
        r = (w.rd64(ins.ops[1]) << (w.rd32(ins.ops[2]) & U32(7)).astype(U64)) + w.rd64(ins.ops[3])
    elif b == "v_pk_mov_b32":
        x, y = w.rd64(ins.ops[1]), w.rd64(ins.ops[2])
        osel = [int(t) for t in ins.mods.get("op_sel", "[0,0]").strip("[]").split(",")]
        lo = (x >> U64(32 * osel[0])) & U64(M32)
        hi = (y >> U64(32 * osel[1])) & U64(M32)
        r = lo | (hi << U64(32))
    else:
        raise Unknown(ins.text)
    w.wr64(ins.ops[0], r)


def ex_mbcnt(w, ins):
    m = w.rd_s(ins.ops[1]) if ins.ops[1].kind != "v" else None
    if m is None:
        raise Unknown(ins.text)
    add = w.rd32(ins.ops[2])
    r = np.zeros(64, dtype=U32)
    hi = ins.base == "v_mbcnt_hi_u32_b32"
    for l in range(64):
        k = (l - 32 if hi else l)
        k = max(0, min(32, k))
        r[l] = bin(m & ((1 << k) - 1)).count("1")
    w.wr32(ins.ops[0], r + add)


def ex_readlane(w, ins):
    w.chk_v(ins.ops[1].n)
    lane = w.rd_s(ins.ops[2]) & 63
    w.sset(ins.ops[0].n, int(w.v[ins.ops[1].n][lane]))


def ex_readfirstlane(w, ins):
    e = w.exec
    lane = (e & -e).bit_length() - 1 if e else 0
    if ins.ops[1].kind == "v":
        w.chk_v(ins.ops[1].n)
        w.sset(ins.ops[0].n, int(w.v[ins.ops[1].n][lane]))
    else:
        w.sset(ins.ops[0].n, w.rd_s(ins.ops[1]))


def ex_writelane(w, ins):
    n = ins.ops[0].n
    if n in w.pv:
        raise Hazard("v_writelane into an outstanding register, line %d" % ins.line)
    w.v[n][w.rd_s(ins.ops[2]) & 63] = w.rd_s(ins.ops[1])


def ex_permlane_swap(w, ins):
    a, b = ins.ops[0].n, ins.ops[1].n
    w.chk_v(a)
    w.chk_v(b)
    if w.exec != 0xFFFFFFFFFFFFFFFF:
        raise Unknown("v_permlane*_swap under a partial EXEC (line %d)" % ins.line)
    va, vb = w.v[a].copy(), w.v[b].copy()
    if ins.base == "v_permlane32_swap_b32":
        ia, ib = np.arange(32, 64), np.arange(0, 32)
    else:                                               # odd rows of the first operand <-> even rows of the second
        ia = np.concatenate([np.arange(16, 32), np.arange(48, 64)])
        ib = np.concatenate([np.arange(0, 16), np.arange(32, 48)])
    w.v[a][ia] = vb[ib]
    w.v[b][ib] = va[ia]


# ------------------------------------------------------------------------------------------------------------ SALU
def sx(v):
    return v - (1 << 32) if v & 0x80000000 else v


def sx64(v):
    return v - (1 << 64) if v & (1 << 63) else v


def simm16(op):
    v = op.val & 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def ex_salu(w, ins):
    b, o = ins.base, ins.ops
    g, g64 = w.rd_s, w.rd_s64
    if b == "s_mov_b32":
        w.sset(o[0].n, g(o[1]))
    elif b == "s_mov_b64":
        w.sset64(o[0].n, g64(o[1]))
    elif b == "s_movk_i32":
        w.sset(o[0].n, simm16(o[1]))
    elif b == "s_not_b32":
        r = ~g(o[1]) & M32
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b == "s_not_b64":
        r = ~g64(o[1]) & 0xFFFFFFFFFFFFFFFF
        w.sset64(o[0].n, r)
        w.scc = int(r != 0)
    elif b in ("s_add_u32", "s_addc_u32"):
        r = g(o[1]) + g(o[2]) + (w.scc if b == "s_addc_u32" else 0)
        w.sset(o[0].n, r)
        w.scc = int(r > M32)
    elif b in ("s_sub_u32", "s_subb_u32"):
        r = g(o[1]) - g(o[2]) - (w.scc if b == "s_subb_u32" else 0)
        w.sset(o[0].n, r)
        w.scc = int(r < 0)
    elif b == "s_add_i32":
        x, y = sx(g(o[1])), sx(g(o[2]))
        r = x + y
        w.sset(o[0].n, r)
        w.scc = int(not (-(1 << 31) <= r < (1 << 31)))
    elif b == "s_sub_i32":
        x, y = sx(g(o[1])), sx(g(o[2]))
        r = x - y
        w.sset(o[0].n, r)
        w.scc = int(not (-(1 << 31) <= r < (1 << 31)))
    elif b == "s_addk_i32":
        r = sx(g(o[0])) + simm16(o[1])
        w.sset(o[0].n, r)
        w.scc = int(not (-(1 << 31) <= r < (1 << 31)))
    elif b == "s_mulk_i32":
        w.sset(o[0].n, sx(g(o[0])) * simm16(o[1]))
    elif b == "s_mul_i32":
        w.sset(o[0].n, sx(g(o[1])) * sx(g(o[2])))
    elif b == "s_mul_hi_u32":
        w.sset(o[0].n, (g(o[1]) * g(o[2])) >> 32)
    elif b == "s_mul_hi_i32":
        w.sset(o[0].n, (sx(g(o[1])) * sx(g(o[2]))) >> 32)
    elif b in ("s_and_b32", "s_or_b32", "s_xor_b32", "s_andn2_b32", "s_orn2_b32"):
        x, y = g(o[1]), g(o[2])
        r = {"s_and_b32": x & y, "s_or_b32": x | y, "s_xor_b32": x ^ y, "s_andn2_b32": x & ~y, "s_orn2_b32": x | ~y}[b] & M32
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b in ("s_and_b64", "s_or_b64", "s_xor_b64", "s_andn2_b64", "s_orn2_b64"):
        x, y = g64(o[1]), g64(o[2])
        r = {"s_and_b64": x & y, "s_or_b64": x | y, "s_xor_b64": x ^ y, "s_andn2_b64": x & ~y, "s_orn2_b64": x | ~y}[b] & 0xFFFFFFFFFFFFFFFF
        w.sset64(o[0].n, r)
        w.scc = int(r != 0)
    elif b in ("s_lshl_b32", "s_lshr_b32", "s_ashr_i32"):
        x, n = g(o[1]), g(o[2]) & 31
        r = (x << n) & M32 if b == "s_lshl_b32" else (x >> n if b == "s_lshr_b32" else (sx(x) >> n) & M32)
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b in ("s_lshl_b64", "s_lshr_b64"):
        x, n = g64(o[1]), g(o[2]) & 63
        r = (x << n) & 0xFFFFFFFFFFFFFFFF if b == "s_lshl_b64" else x >> n
        w.sset64(o[0].n, r)
        w.scc = int(r != 0)
    elif b in ("s_min_i32", "s_max_i32"):
        x, y = sx(g(o[1])), sx(g(o[2]))
        first = x <= y if b == "s_min_i32" else x >= y
        w.sset(o[0].n, x if first else y)
        w.scc = int(first)
    elif b == "s_abs_i32":
        r = abs(sx(g(o[1]))) & M32
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b == "s_cselect_b32":
        w.sset(o[0].n, g(o[1]) if w.scc else g(o[2]))
    elif b == "s_cselect_b64":
        w.sset64(o[0].n, g64(o[1]) if w.scc else g64(o[2]))
    elif b.startswith("s_cmp_") or b.startswith("s_cmpk_"):
        p = b.split("_")
        op, ty = p[2], p[3]
        if b.startswith("s_cmpk_"):
            x = g(o[0])
            y = simm16(o[1]) if ty == "i32" else o[1].val & 0xFFFF
            if ty == "i32":
                x = sx(x)
        elif ty == "u64":
            x, y = g64(o[0]), g64(o[1])
        else:
            x, y = g(o[0]), g(o[1])
            if ty == "i32":
                x, y = sx(x), sx(y)
        w.scc = int(CMP[op](x, y))
    elif b in ("s_bitcmp0_b32", "s_bitcmp1_b32"):
        bit = (g(o[0]) >> (g(o[1]) & 31)) & 1
        w.scc = int(bit == (1 if b == "s_bitcmp1_b32" else 0))
    elif b == "s_bcnt1_i32_b64":
        r = bin(g64(o[1])).count("1")
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b == "s_bcnt1_i32_b32":
        r = bin(g(o[1])).count("1")
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b in ("s_ff1_i32_b32", "s_ff1_i32_b64"):
        x = g(o[1]) if b.endswith("b32") else g64(o[1])
        w.sset(o[0].n, (x & -x).bit_length() - 1 if x else M32)
    elif b in ("s_flbit_i32_b32", "s_flbit_i32_b64"):
        bits = 32 if b.endswith("b32") else 64
        x = g(o[1]) if bits == 32 else g64(o[1])
        w.sset(o[0].n, bits - x.bit_length() if x else M32)
    elif b == "s_bfe_u32":
        x, c = g(o[1]), g(o[2])
        off, wd = c & 31, (c >> 16) & 0x7F
        r = (x >> off) & ((1 << wd) - 1) if wd else 0
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b == "s_ashr_i64":
        r = (sx64(g64(o[1])) >> (g(o[2]) & 63)) & 0xFFFFFFFFFFFFFFFF
        w.sset64(o[0].n, r)
        w.scc = int(r != 0)
    elif b == "s_bfe_i32":
        x, c = g(o[1]), g(o[2])
        off, wd = c & 31, (c >> 16) & 0x7F
        r = (x >> off) & ((1 << wd) - 1) if wd else 0
        if wd and wd < 32 and r & (1 << (wd - 1)):
            r |= M32 & ~((1 << wd) - 1)
        w.sset(o[0].n, r)
        w.scc = int(r != 0)
    elif b == "s_bfe_i64":
        x, c = g64(o[1]), g(o[2])
        off, wd = c & 63, (c >> 16) & 0x7F
        r = (x >> off) & ((1 << wd) - 1) if wd else 0
        if wd and wd < 64 and r & (1 << (wd - 1)):
            r |= 0xFFFFFFFFFFFFFFFF & ~((1 << wd) - 1)
        w.sset64(o[0].n, r)
        w.scc = int(r != 0)
    elif b in ("s_bitset0_b32", "s_bitset1_b32"):
        bit = 1 << (g(o[1]) & 31)
        w.sset(o[0].n, (g(o[0]) | bit) if b == "s_bitset1_b32" else (g(o[0]) & ~bit))
    elif b == "s_brev_b32":
        w.sset(o[0].n, int("{:032b}".format(g(o[1]))[::-1], 2))
    elif b in ("s_min_u32", "s_max_u32"):
        x, y = g(o[1]), g(o[2])
        first = x <= y if b == "s_min_u32" else x >= y
        w.sset(o[0].n, x if first else y)
        w.scc = int(first)
    elif b == "s_nor_b64":
        r = ~(g64(o[1]) | g64(o[2])) & 0xFFFFFFFFFFFFFFFF
        w.sset64(o[0].n, r)
        w.scc = int(r != 0)
    elif b == "s_pack_ll_b32_b16":
        w.sset(o[0].n, (g(o[1]) & 0xFFFF) | ((g(o[2]) & 0xFFFF) << 16))
    elif b in ("s_and_saveexec_b64", "s_or_saveexec_b64", "s_andn2_saveexec_b64"):
        x, e = g64(o[1]), w.exec
        w.sset64(o[0].n, e)
        r = {"s_and_saveexec_b64": x & e, "s_or_saveexec_b64": x | e, "s_andn2_saveexec_b64": x & ~e}[b] & 0xFFFFFFFFFFFFFFFF
        w.sset64(EXEC, r)
        w.scc = int(r != 0)
    else:
        raise Unknown(ins.text)
    if o and o[0].kind == "s" and w.ps:
        for k in range(o[0].n, o[0].n + o[0].cnt):
            if k in w.ps and not b.startswith("s_cmp") and not b.startswith("s_bitcmp"):
                raise Hazard("s%d overwritten at line %d while its load (line %d) is outstanding" % (k, ins.line, w.ps[k]))


def ex_branch(w, ins):
    b = ins.base
    take = {"s_branch": True, "s_cbranch_scc0": w.scc == 0, "s_cbranch_scc1": w.scc == 1, "s_cbranch_vccz": w.sget64(VCC) == 0,
            "s_cbranch_vccnz": w.sget64(VCC) != 0, "s_cbranch_execz": w.exec == 0, "s_cbranch_execnz": w.exec != 0}[b]
    if take:
        w.pc = w.labels[ins.ops[0].text]


def ex_nop(w, ins):
    return None


def ex_unknown(w, ins):
    raise Unknown("no semantics for: %s (line %d)" % (ins.text, ins.line))


def ex_call(w, ins):
    b = ins.base
    nxt = CODE_BASE + 4 * w.pc                       # (pc already points behind this instruction)
    if b == "s_getpc_b64":
        w.sset64(ins.ops[0].n, nxt)
        w.getpc = nxt
        return
    tgt = w.rd_s64(ins.ops[1] if b == "s_swappc_b64" else ins.ops[0])
    k = tgt - CODE_BASE
    if k < 0 or k & 3 or (k >> 2) >= len(w.prog):
        raise Hazard("jump to %#x at line %d (%s)" % (tgt, ins.line, ins.text))
    if b == "s_swappc_b64":
        w.sset64(ins.ops[0].n, nxt)
    w.pc = k >> 2


def ex_waitcnt(w, ins):
    w.waitcnt(ins)


def ex_barrier(w, ins):
    return "barrier"


def ex_endpgm(w, ins):
    w.done = True
    return "end"


def ex_gpr_idx_on(w, ins):
    mode = ins.mods.get("gpr_idx")
    if mode not in ("DST", "SRC0"):
        raise Unknown(ins.text)
    idx = w.rd_s(ins.ops[0]) & 0xFF
    w.sset(M0, (w.sget(M0) & ~0xF0FF) | idx | ((8 if mode == "DST" else 1) << 12))
    w.idx_dst = idx if mode == "DST" else None
    w.idx_src0 = idx if mode == "SRC0" else None


def ex_gpr_idx_off(w, ins):
    w.idx_dst = None
    w.idx_src0 = None


# ------------------------------------------------------------------------------------------------------------ memory instructions
def off_of(ins, key="offset"):
    return int(ins.mods.get(key, "0"), 0)


def ex_sload(w, ins):
    n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[ins.base]
    base = w.rd_s64(ins.ops[1])
    o = ins.ops[2]
    addr = (base + (w.rd_s(o) if o.kind == "s" else o.val)) & ~3
    raw = w.mem.read_scalar(addr, 4 * n)
    d = ins.ops[0].n
    regs = []
    for k in range(n):
        if d + k in w.ps:
            raise Hazard("s%d reloaded while outstanding (line %d)" % (d + k, ins.line))
        w.sset(d + k, struct.unpack_from("<I", raw, 4 * k)[0])
        regs.append(("s", d + k))
    w.poison(w.lgkm, "smem", regs)


GL_BYTES = {"ubyte": 1, "sbyte": 1, "ushort": 2, "sshort": 2, "dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "byte": 1, "short": 2}


def gaddr(w, ins, vaddr, saddr):
    off = off_of(ins)
    if off & 0x1000 and off < 0x2000 and "-" not in ins.mods.get("offset", ""):
        pass
    if saddr.kind == "off":
        a = w.rd64(vaddr)
    else:
        a = w.rd32(vaddr).astype(U64) + U64(w.rd_s64(saddr))
    return (a.view(I64) + I64(int(ins.mods.get("offset", "0"), 0))).view(U64)


def ex_gload(w, ins):
    ty = ins.base[ins.base.index("_load_") + 6:]
    d16 = None
    if "_d16" in ty:                                     # 16 bits of the destination are written, the other half is kept
        d16 = "hi" if ty.endswith("_d16_hi") else "lo"
        ty = ty[:ty.index("_d16")]
    nb = GL_BYTES[ty]
    if ins.base.startswith("flat_"):                    # (a flat address of these kernels is a global one: no LDS / scratch apertures in use)
        addr = (w.rd64(ins.ops[1]).view(I64) + I64(int(ins.mods.get("offset", "0"), 0))).view(U64)
    else:
        addr = gaddr(w, ins, ins.ops[1], ins.ops[2])
    m = w.execm()
    raw = w.mem.read(addr, nb, m, ins.text)
    d = ins.ops[0].n
    regs = []
    nd = max(1, nb // 4)
    for k in range(nd):
        if d + k in w.pv:
            raise Hazard("v%d reloaded while outstanding (line %d)" % (d + k, ins.line))
        chunk = raw[:, 4 * k:4 * k + 4]
        val = np.zeros(64, dtype=U32)
        for q in range(chunk.shape[1]):
            val |= chunk[:, q].astype(U32) << U32(8 * q)
        if ty == "sbyte":
            val = (val.astype(np.uint8).view(np.int8).astype(I32)).view(U32)
        if ty == "sshort":
            val = (val.astype(np.uint16).view(np.int16).astype(I32)).view(U32)
        if d16 == "hi":
            val = (w.v[d + k] & U32(0xFFFF)) | ((val & U32(0xFFFF)) << U32(16))
        elif d16 == "lo":
            val = (w.v[d + k] & U32(0xFFFF0000)) | (val & U32(0xFFFF))
        w.v[d + k] = np.where(m, val, w.v[d + k])
        regs.append(("v", d + k))
    w.poison(w.vm, "load", regs)
    if ins.base.startswith("flat_"):
        w.lgkm.append(("lds", []))                   # (a flat access counts on both counters)


def vbytes(w, op, nb):
    nd = max(1, (nb + 3) // 4)
    w.chk_v(op.n, nd)
    out = np.zeros((64, 4 * nd), dtype=np.uint8)
    for k in range(nd):
        for q in range(4):
            out[:, 4 * k + q] = (w.v[op.n + k] >> U32(8 * q)).astype(np.uint8)
    return out[:, :nb]


def ex_gstore(w, ins):
    ty = ins.base[len("global_store_"):]
    hi = ty.endswith("_d16_hi")                          # the upper 16 bits of the register are what is stored
    if hi:
        ty = ty[:-7]
    nb = GL_BYTES[ty]
    addr = gaddr(w, ins, ins.ops[0], ins.ops[2])
    data = vbytes(w, ins.ops[1], 4)[:, 2:2 + nb] if hi else vbytes(w, ins.ops[1], nb)
    w.mem.write(addr, data, w.execm(), ins.text)
    w.vm.append(("store", []))


def scratch_addr(w, ins, vaddr, saddr):
    a = np.zeros(64, dtype=np.int64)
    if vaddr.kind == "v":
        a = a + w.rd32(vaddr).astype(np.int64)
    if saddr.kind == "s":
        a = a + w.rd_s(saddr)
    return a + int(ins.mods.get("offset", "0"), 0)


def ex_scratch(w, ins):
    """the wave's private segment (architected flat scratch): lane l's bytes at scratch[l, address]"""
    load = ins.base.startswith("scratch_load_")
    nb = GL_BYTES[ins.base[len("scratch_load_" if load else "scratch_store_"):]]
    m = w.execm()
    if load:
        a = scratch_addr(w, ins, ins.ops[1], ins.ops[2])
    else:
        a = scratch_addr(w, ins, ins.ops[0], ins.ops[2])
    if m.any() and (int(a[m].min()) < 0 or int(a[m].max()) + nb > w.scratch.shape[1]):
        raise Hazard("scratch access outside the private segment at line %d (%s)" % (ins.line, ins.text))
    idx = a[:, None] + np.arange(nb)[None, :]
    if load:
        raw = np.zeros((64, nb), dtype=np.uint8)
        raw[m] = w.scratch[LANES[m][:, None], idx[m]]
        d = ins.ops[0].n
        regs = []
        for k in range(max(1, nb // 4)):
            if d + k in w.pv:
                raise Hazard("v%d reloaded while outstanding (line %d)" % (d + k, ins.line))
            w.v[d + k] = np.where(m, words(raw, k), w.v[d + k])
            regs.append(("v", d + k))
        w.poison(w.vm, "load", regs)
    else:
        data = vbytes(w, ins.ops[1], nb)
        w.scratch[LANES[m][:, None], idx[m]] = data[m]
        w.vm.append(("store", []))


def ex_gatomic(w, ins):
    b = ins.base[len("global_atomic_"):]
    wide = b.endswith("_x2")
    op = b[:-3] if wide else b
    ret = "sc0" in ins.flags or "glc" in ins.flags
    ops = ins.ops
    if ret:
        dst, vaddr, vdata, saddr = ops
    else:
        dst = None
        vaddr, vdata, saddr = ops
    nb = 8 if wide else 4
    addr = gaddr(w, ins, vaddr, saddr)
    cmp = None
    if op == "cmpswap":                     # DATA[0] = the value to store, DATA[1] = the value to compare with (two / four registers)
        half = Op("v", vdata.n, vdata.cnt // 2, text="v%d" % vdata.n)
        other = Op("v", vdata.n + vdata.cnt // 2, vdata.cnt // 2, text="v%d" % (vdata.n + vdata.cnt // 2))
        data = w.rd64(half) if wide else w.rd32(half).astype(U64)
        cmp = w.rd64(other) if wide else w.rd32(other).astype(U64)
    else:
        data = w.rd64(vdata) if wide else w.rd32(vdata).astype(U64)
    m = w.execm()
    old_out = np.zeros(64, dtype=U64)
    fmt = "<Q" if wide else "<I"
    mask = 0xFFFFFFFFFFFFFFFF if wide else M32
    for l in np.nonzero(m)[0]:
        x = int(addr[l])
        j = w.mem.find(x, x + nb, ins.text)
        buf = w.mem.bufs[j]
        o = x - w.mem.bases[j]
        old = struct.unpack_from(fmt, buf, o)[0]
        d = int(data[l])
        if cmp is not None:
            new = d if old == int(cmp[l]) else old
        else:
            new = {"add": (old + d) & mask, "umin": min(old, d), "umax": max(old, d), "swap": d, "or": old | d, "and": old & d}[op]
        struct.pack_into(fmt, buf, o, new)
        old_out[l] = old
    if ret:
        if wide:
            w.wr64(dst, old_out)
        else:
            w.wr32(dst, old_out.astype(U32))
        w.poison(w.vm, "load", [("v", dst.n + k) for k in range(2 if wide else 1)])
    else:
        w.vm.append(("store", []))


def ex_dma(w, ins):
    """global_load_lds_dwordx4 vaddr, saddr|off: lane l's 16 bytes land at LDS[M0 + offset + 16 l] (active lanes only)"""
    addr = gaddr(w, ins, ins.ops[0], ins.ops[1])
    m = w.execm()
    raw = w.mem.read(addr, 16, m, ins.text)
    base = w.sget(M0) + int(ins.mods.get("offset", "0"), 0)
    lds = w.wg.lds
    idx = (base + 16 * LANES[m])[:, None] + np.arange(16)[None, :]
    lds[idx] = raw[m]
    w.wg.lds_poison[idx] += 1
    w.vm.append(("dma", idx))


def lds_rd(w, addrs, nb, m, ins):
    lds = w.wg.lds
    out = np.zeros((64, nb), dtype=np.uint8)
    a = addrs.astype(np.int64)
    if m.any():
        if int(a[m].max()) + nb > lds.size:
            raise Hazard("LDS read past the allocation at line %d (%s)" % (ins.line, ins.text))
        idx = a[m][:, None] + np.arange(nb)[None, :]
        got = lds[idx]
        racy = w.wg.lds_poison[idx] != 0
        if racy.any():
            # bytes of a DMA that is still in flight: the hardware returns old or new bytes.  Not an error by itself (lanes past a
            # row's pitch over-read into the next buffer and never use what they get) -- the read returns GARBAGE here, so a result
            # that depends on it no longer matches the emulated twin's
            got = np.where(racy, np.uint8(0xD7), got)
            w.wg.racy_reads += int(racy.sum())
            w.wg.racy_lines.add(ins.line)
        out[m] = got
    return out


def words(raw, k):
    c = raw[:, 4 * k:4 * k + 4]
    v = np.zeros(64, dtype=U32)
    for q in range(c.shape[1]):
        v |= c[:, q].astype(U32) << U32(8 * q)
    return v


def ex_ds(w, ins):
    b, o = ins.base, ins.ops
    m = w.execm()
    lds = w.wg.lds
    if b in ("ds_read_b32", "ds_read_b64", "ds_read_b96", "ds_read_b128", "ds_read_u8", "ds_read_u16", "ds_read_i8"):
        nb = {"b32": 4, "b64": 8, "b96": 12, "b128": 16, "u8": 1, "u16": 2, "i8": 1}[b[8:]]
        addr = w.rd32(o[1]) + U32(off_of(ins))
        raw = lds_rd(w, addr, nb, m, ins)
        regs = []
        for k in range(max(1, nb // 4)):
            if o[0].n + k in w.pv:
                raise Hazard("v%d reloaded while outstanding (line %d)" % (o[0].n + k, ins.line))
            w.v[o[0].n + k] = np.where(m, words(raw, k), w.v[o[0].n + k])
            regs.append(("v", o[0].n + k))
        w.poison(w.lgkm, "lds", regs)
    elif b in ("ds_read2_b32", "ds_read2st64_b64", "ds_read2_b64", "ds_read2st64_b32"):
        el = 8 if b.endswith("b64") else 4
        stride = el * (64 if "st64" in b else 1)
        base = w.rd32(o[1])
        regs = []
        for h, key in enumerate(("offset0", "offset1")):
            addr = base + U32(off_of(ins, key) * stride)
            raw = lds_rd(w, addr, el, m, ins)
            for k in range(el // 4):
                r = o[0].n + h * (el // 4) + k
                w.v[r] = np.where(m, words(raw, k), w.v[r])
                regs.append(("v", r))
        w.poison(w.lgkm, "lds", regs)
    elif b in ("ds_write_b32", "ds_write_b64", "ds_write_b128", "ds_write_b8", "ds_write_b16", "ds_write_b96"):
        nb = {"b32": 4, "b64": 8, "b96": 12, "b128": 16, "b8": 1, "b16": 2}[b[9:]]
        addr = (w.rd32(o[0]) + U32(off_of(ins))).astype(np.int64)
        data = vbytes(w, o[1], nb)
        for l in np.nonzero(m)[0]:
            a = int(addr[l])
            if w.wg.lds_poison[a:a + nb].any():
                raise Hazard("LDS bytes of an outstanding DMA are written at line %d" % ins.line)
            lds[a:a + nb] = data[l]
        w.lgkm.append(("lds", []))
    elif b in ("ds_write2_b32", "ds_write2st64_b32", "ds_write2_b64", "ds_write2st64_b64"):
        el = 8 if b.endswith("b64") else 4
        stride = el * (64 if "st64" in b else 1)
        base = w.rd32(o[0]).astype(np.int64)
        for h, key in enumerate(("offset0", "offset1")):
            addr = base + off_of(ins, key) * stride
            data = vbytes(w, o[1 + h], el)
            for l in np.nonzero(m)[0]:
                a = int(addr[l])
                if w.wg.lds_poison[a:a + el].any():
                    raise Hazard("LDS bytes of an outstanding DMA are written at line %d" % ins.line)
                lds[a:a + el] = data[l]
        w.lgkm.append(("lds", []))
    elif b in ("ds_add_u32", "ds_or_b32", "ds_max_u32", "ds_min_u32"):
        addr = (w.rd32(o[0]) + U32(off_of(ins))).astype(np.int64)
        data = w.rd32(o[1])
        l32 = lds.view(U32)
        for l in np.nonzero(m)[0]:
            a = int(addr[l])
            if a & 3:
                raise Hazard("unaligned LDS atomic at line %d" % ins.line)
            if w.wg.lds_poison[a:a + 4].any():
                raise Hazard("LDS atomic on bytes of an outstanding DMA at line %d" % ins.line)
            old, d = int(l32[a >> 2]), int(data[l])
            l32[a >> 2] = {"ds_add_u32": (old + d) & M32, "ds_or_b32": old | d, "ds_max_u32": max(old, d), "ds_min_u32": min(old, d)}[b]
        w.lgkm.append(("lds", []))
    elif b in ("ds_max_u64", "ds_min_u64", "ds_add_u64"):
        addr = (w.rd32(o[0]) + U32(off_of(ins))).astype(np.int64)
        data = w.rd64(o[1])
        l64 = lds.view(U64)
        for l in np.nonzero(m)[0]:
            a = int(addr[l])
            old, d = int(l64[a >> 3]), int(data[l])
            l64[a >> 3] = {"ds_max_u64": max(old, d), "ds_min_u64": min(old, d), "ds_add_u64": (old + d) & 0xFFFFFFFFFFFFFFFF}[b]
        w.lgkm.append(("lds", []))
    elif b == "ds_bpermute_b32":
        addr = w.rd32(o[1]) + U32(off_of(ins))
        data = w.rd32(o[2])
        sl = ((addr >> U32(2)) & U32(63)).astype(np.int64)
        val = np.where(m[sl], data[sl], U32(0))
        if o[0].n in w.pv:
            raise Hazard("v%d reloaded while outstanding (line %d)" % (o[0].n, ins.line))
        w.v[o[0].n] = np.where(m, val, w.v[o[0].n])
        w.poison(w.lgkm, "lds", [("v", o[0].n)])
    else:
        raise Unknown(ins.text)


# ------------------------------------------------------------------------------------------------------------ dispatch
def handler(ins):
    b = ins.base
    if b in VOP:
        return ex_valu
    if b in ("v_bitop3_b32", "v_bitop3_b16"):
        return ex_bitop3
    if b == "v_fmac_f32":
        return ex_fmac
    if b.startswith("v_cmp_"):
        return ex_vcmp
    if b == "v_cndmask_b32":
        return ex_cndmask
    if b in ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_subb_co_u32", "v_addc_co_u32", "v_subbrev_co_u32"):
        return ex_carry
    if b in ("v_mad_u64_u32", "v_mad_i64_i32"):
        return ex_mad64
    if b in ("v_mov_b64", "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64", "v_pk_mov_b32"):
        return ex_v64
    if b in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
        return ex_mbcnt
    if b == "v_readlane_b32":
        return ex_readlane
    if b == "v_readfirstlane_b32":
        return ex_readfirstlane
    if b == "v_writelane_b32":
        return ex_writelane
    if b in ("v_permlane16_swap_b32", "v_permlane32_swap_b32"):
        return ex_permlane_swap
    if b in ("s_getpc_b64", "s_swappc_b64", "s_setpc_b64"):
        return ex_call
    if b in ("s_branch",) or b.startswith("s_cbranch_"):
        return ex_branch
    if b in ("s_nop", "s_setprio", "buffer_wbl2", "buffer_inv", "s_sleep", "s_sethalt"):
        return ex_nop
    if b == "s_waitcnt":
        return ex_waitcnt
    if b == "s_barrier":
        return ex_barrier
    if b == "s_endpgm":
        return ex_endpgm
    if b == "s_set_gpr_idx_on":
        return ex_gpr_idx_on
    if b == "s_set_gpr_idx_off":
        return ex_gpr_idx_off
    if b.startswith("s_load_"):
        return ex_sload
    if b == "global_load_lds_dwordx4":
        return ex_dma
    if b.startswith("scratch_load_") or b.startswith("scratch_store_"):
        return ex_scratch
    if b.startswith("global_load_") or b.startswith("flat_load_"):
        return ex_gload
    if b.startswith("global_store_"):
        return ex_gstore
    if b.startswith("global_atomic_"):
        return ex_gatomic
    if b.startswith("ds_"):
        return ex_ds
    if b.startswith("s_"):
        return ex_salu
    raise Unknown(ins.text)


def bind(prog):
    missing = {}
    for ins in prog:
        try:
            ins.fn = handler(ins)
        except Unknown:
            missing[ins.base] = missing.get(ins.base, 0) + 1
    if missing:
        raise Unknown("no semantics for: " + ", ".join("%s (%d)" % kv for kv in sorted(missing.items())))


# ------------------------------------------------------------------------------------------------------------ launch
class Workgroup:
    def __init__(self, lds_bytes):
        self.lds = np.zeros(lds_bytes + 4096, dtype=np.uint8)          # (+ slack: the kernel's own over-read margin is inside its allocation)
        self.lds[:] = 0xEE
        self.lds_poison = np.zeros(lds_bytes + 4096, dtype=np.uint8)
        self.racy_reads = 0
        self.racy_lines = set()


KERNARG_BASE = 0x7E0000000000


def run_launch(asm_path, symbol, kernarg, grid, block, shmem, mem, workgroups=None, trace=None, progress=None, schedule=None, garbage=None, coverage=None,
               max_wave_instructions=None):
    """runs the workgroups (all of them by default) one after the other; -> {instructions, hazards: [...]}

    schedule: how the wavefronts of a workgroup take turns between two barriers (the hardware promises no order at all):
      None / "forward"   wave 0 up to its barrier (or its end), then wave 1, ...               (the emulated twin's order)
      "reverse"          the last wave first, the workgroups from the last one down
      "random:SEED[:Q]"  pre-emptive: a random runnable wave runs 1..Q (default 40) instructions, then the next draw; the
                         workgroups in a shuffled order.
      "skew:SEED[:Q]"    the same with a pace per wave and barrier interval (turns of at most 2, Q or 10 Q instructions): one wave
                         can still be inside a loop the others left long ago.  A kernel whose waves hand data to each other through LDS or memory
                         without a barrier in between leaves different memory under one of these orders -- unless it means to
                         (a slot counter, first-come-first-kept tables): those kernels are named by their tests."""
    prog, labels, kd = parse_function(asm_path, symbol)
    kernarg = kernarg_segment(asm_path, symbol, kernarg, grid, block, shmem)
    if kd.get("user_sgpr_count", 2) != 2 or not kd.get("user_sgpr_kernarg_segment_ptr", 1):
        raise Unknown("kernel ABI other than {kernarg pointer, workgroup id x}")
    if KERNARG_BASE not in mem.bases:
        mem.add(KERNARG_BASE, bytes(kernarg) + bytes(256))
    data_syms, image = parse_data(asm_path)
    if image and DATA_BASE not in mem.bases:
        mem.add(DATA_BASE, image + bytes(64))
    if mem.symbol_address is None:
        mem.symbol_address = lambda name: DATA_BASE + data_syms[name] if name in data_syms else None
    static_lds = kd.get("group_segment_fixed_size", 0)
    if static_lds + shmem > 160 * 1024:
        raise Hazard("%d + %d bytes of LDS (static + dynamic): more than a CU has" % (static_lds, shmem))
    nthreads = block[0] * block[1] * block[2]
    total, racy, racy_lines = 0, 0, set()
    wg_order = list(range(grid[0]) if workgroups is None else workgroups)
    rng, quantum = None, 40
    if schedule == "reverse":
        wg_order.reverse()
    elif schedule and schedule.split(":")[0] in ("random", "skew"):
        import random
        parts = schedule.split(":")
        rng = random.Random(int(parts[1]) if len(parts) > 1 else 1)
        quantum = int(parts[2]) if len(parts) > 2 else 40
        skew = parts[0] == "skew"
        rng.shuffle(wg_order)
    elif schedule not in (None, "forward"):
        raise ValueError("schedule: " + schedule)
    for wgx in wg_order:
        wg = Workgroup(static_lds + shmem)
        waves = []
        for wi in range((nthreads + 63) // 64):
            w = Wave(wg, wi, prog, labels, mem, kd)
            w.data_syms = data_syms
            w.pc = labels[symbol]
            if garbage is not None:
                # what a wave finds in its registers is what the wave before it left there: every VGPR, every SGPR but the ones the
                # kernel descriptor asks for, VCC, M0 and SCC start as noise (seeded) -- code that counts on a zero it never wrote
                # (a lane an EXEC-masked write skipped, a v_writelane into a register nobody initialised) leaves other memory
                g = np.random.default_rng([int(garbage), wgx, wi])
                w.v[:] = g.integers(0, 1 << 32, size=w.v.shape, dtype=np.uint64).astype(U32)
                w.s[:106] = g.integers(0, 1 << 32, size=106, dtype=np.uint64)
                w.s[VCC:VCC + 2] = g.integers(0, 1 << 32, size=2, dtype=np.uint64)
                w.s[M0] = int(g.integers(0, 1 << 32))
                w.scc = int(g.integers(0, 2))
            w.sset64(0, KERNARG_BASE)
            w.sset(2, wgx)
            tid = wi * 64 + LANES
            live = tid < nthreads
            w.v[0] = np.where(live, tid % block[0], 0).astype(U32)          # (one-dimensional blocks: y = z = 0 in bits 10.., 20..)
            ex = 0
            for l in range(64):
                if live[l]:
                    ex |= 1 << l
            w.sset64(EXEC, ex)
            w.trace = trace
            w.cov = coverage                       # (a numpy array of len(program of the file): executed-instruction counters, all waves)
            w.cap = max_wave_instructions
            waves.append(w)
        live = list(waves)
        if rng is not None:
            waiting = []
            fresh = True
            while live or waiting:
                if not live:                       # every wave that has not ended stands at the barrier
                    live, waiting = waiting, []
                    fresh = True
                    continue
                if fresh:                          # "skew": between two barriers every wave has a pace of its own
                    for w in live:
                        w.qcap = rng.choice((2, quantum, 10 * quantum)) if skew else quantum
                    fresh = False
                w = live[rng.randrange(len(live))]
                r = w.run(limit=w.nexec + rng.randint(1, w.qcap))
                if r == "barrier":
                    live.remove(w)
                    waiting.append(w)
                elif r == "end":
                    live.remove(w)
        while live:
            at_barrier = []
            for w in (reversed(live) if schedule == "reverse" else live):
                r = w.run()
                if r == "barrier":
                    at_barrier.append(w)
            live = at_barrier[::-1] if schedule == "reverse" else at_barrier
        for w in waves:
            total += w.nexec
        racy += wg.racy_reads
        racy_lines |= wg.racy_lines
        if progress:
            progress(wgx, total)
    both = sorted(mem.scalar_lines & mem.written_lines)
    return {"instructions": total, "lds_bytes_read_under_a_dma": racy, "at_lines": sorted(racy_lines),
            "scalar_lines_also_written": len(both), "first_such_line": ("%#x" % (both[0] << 6)) if both else None,
            "scalar_loads_of_words_written_in_this_launch": len(mem.stale_scalar_reads)}


# ------------------------------------------------------------------------------------------------------------ captured launches
def load_dump(dump_dir, k):
    meta = json.load(open(os.path.join(dump_dir, "L%d.json" % k)))
    return meta, read_image(os.path.join(dump_dir, "L%d.pre" % k)), read_image(os.path.join(dump_dir, "L%d.post" % k))


def read_image(path):
    """a snapshot of tests/simt/simt_runtime.cpp's dump_mem: the pieces that are not all 0xEE, at their offsets"""
    raw = open(path, "rb").read()
    if raw[:8] != b"SNKDUMP1":
        return raw
    total = struct.unpack_from("<Q", raw, 8)[0]
    img = bytearray(b"\xee") * total
    o = 16
    while o < len(raw):
        at, n = struct.unpack_from("<QQ", raw, o)
        if n >> 63:                         # one byte value all over the piece
            n &= (1 << 63) - 1
            img[at:at + n] = raw[o + 16:o + 17] * n
            o += 17
        else:
            img[at:at + n] = raw[o + 16:o + 16 + n]
            o += 16 + n
    return bytes(img)


def symbol_at(lib, offset):
    out = subprocess.run(["nm", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    for line in out.split("\n"):
        p = line.split()
        if len(p) == 3 and int(p[0], 16) == offset:
            return p[2]
    raise KeyError("no symbol at %#x in %s" % (offset, lib))


def kernel_offsets(lib, pattern):
    """offsets of the host twins of the kernels whose mangled name contains `pattern` (SIMT_DUMP_OFFSETS)"""
    out = subprocess.run(["nm", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    return {int(p[0], 16): p[2] for p in (l.split() for l in out.split("\n")) if len(p) == 3 and pattern in p[2]}


_COV_SEQ = [0]


def function_extent(asm_path, symbol):
    """-> (first, last + 1) instruction indices of `symbol` in the file's program: up to the next function symbol (a label that is
    not a local .L one)"""
    prog, labels, _ = parse_file(asm_path)
    a = labels[symbol]
    nxt = [i for name, i in labels.items() if i > a and not name.startswith(".L") and not name.startswith("BB")]
    return a, (min(nxt) if nxt else len(prog))


def replay(dump_dir, k, asm_path, workgroups=None, verbose=True, schedule=None, keep_memory=False, garbage=None, coverage=None, max_wave_instructions=None):
    """runs launch k of a dump through the assembly (a file, or the build directory whose kept files are searched for the kernel);
    -> (summary, list of differing (allocation, first offset, count)).
    coverage: True -> info["executed_lines"] = the assembly lines of every instruction some wave executed (the kernel and the
    functions it calls).  With SNK_ISA_COV_DIR in the environment every replay leaves such a record there (cov_<pid>_<n>.json:
    file, kernel symbol, lines) -- tools/isa_coverage.py adds them up per kernel."""
    meta, pre, post = load_dump(dump_dir, k)
    sym = symbol_at(meta["lib"], meta["offset"])
    if os.path.isdir(asm_path):
        asm_path = find_asm(asm_path, sym)
    cov_dir = os.environ.get("SNK_ISA_COV_DIR")
    cov = None
    if coverage or cov_dir:
        cov = np.zeros(len(parse_file(asm_path)[0]), dtype=np.uint32)
    mem = Memory()
    o = 0
    spans = []
    for base, size in meta["allocs"]:
        mem.add(base, pre[o:o + size])
        spans.append((base, size, o))
        o += size
    if "base" in meta:                     # a device variable the host fills (hipMemcpyToSymbol): the emulated library's own copy, if it was captured
        data_syms, _ = parse_data(asm_path)
        lib_syms = {v: k for k, v in kernel_offsets(meta["lib"], "").items()}

        def symbol_address(name):
            if name in lib_syms:
                a = meta["base"] + lib_syms[name]
                i = bisect.bisect_right(mem.bases, a) - 1
                if i >= 0 and a < mem.bases[i] + mem.sizes[i]:
                    return a
            return DATA_BASE + data_syms[name] if name in data_syms else None
        mem.symbol_address = symbol_address
    gone = set()
    if os.path.exists(os.path.join(dump_dir, "L%d.gone" % k)):          # freed while the kernel's snapshots were taken (another host thread)
        gone = {int(x) for x in open(os.path.join(dump_dir, "L%d.gone" % k)).read().split()}
    info = run_launch(asm_path, sym, bytes.fromhex(meta["kernarg"]), meta["grid"], meta["block"], meta["shmem"], mem, workgroups, schedule=schedule, garbage=garbage,
                      coverage=cov, max_wave_instructions=max_wave_instructions, progress=(lambda g, n: print("  workgroup %d done, %d wave instructions so far" % (g, n), flush=True)) if verbose else None)
    diffs = []
    for n, (base, size, o) in enumerate(spans):
        if n in gone:
            continue
        i = mem.bases.index(base)
        want = np.frombuffer(post[o:o + size], dtype=np.uint8)
        ne = np.nonzero(mem.bufs[i] != want)[0]
        if ne.size:
            diffs.append((base, int(ne[0]), int(ne.size), size))
            if keep_memory:                  # (for kernels whose memory is order-dependent by design: the caller compares what matters)
                info.setdefault("differing", []).append((base, mem.bufs[i].copy(), want.copy()))
    info.update(symbol=sym, grid=meta["grid"], block=meta["block"], shmem=meta["shmem"])
    if cov is not None:
        prog = parse_file(asm_path)[0]
        lines = [prog[i].line for i in np.nonzero(cov)[0].tolist()]
        if coverage:
            info["executed_lines"] = lines
        if cov_dir:
            os.makedirs(cov_dir, exist_ok=True)
            _COV_SEQ[0] += 1
            with open(os.path.join(cov_dir, "cov_%d_%d.json" % (os.getpid(), _COV_SEQ[0])), "w") as f:
                json.dump({"asm": os.path.basename(asm_path), "symbol": sym, "lines": lines, "identical": not diffs}, f)
    return info, diffs


def main():
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("dump_dir")
    ap.add_argument("launch", type=int)
    ap.add_argument("asm")
    ap.add_argument("--workgroups", type=str, default=None)
    ap.add_argument("--schedule", type=str, default=None, help="forward (default) | reverse | random:SEED[:QUANTUM] | skew:SEED[:QUANTUM]")
    ap.add_argument("--garbage", type=int, default=None, help="seed: registers start as noise instead of zeros (all but the ABI's)")
    a = ap.parse_args()
    wgs = [int(x) for x in a.workgroups.split(",")] if a.workgroups else None
    info, diffs = replay(a.dump_dir, a.launch, a.asm, wgs, schedule=a.schedule, garbage=a.garbage)
    print(json.dumps(info))
    for d in diffs:
        print("DIFFERS: allocation %#x (%d bytes): %d bytes differ, the first at offset %d" % (d[0], d[3], d[2], d[1]))
    sys.exit(1 if diffs else 0)


if __name__ == "__main__":
    main()
