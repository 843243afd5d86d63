// The chunk decoder of the device inflate (soapnuke_amd/csrc/snk_inflate_core.hip.h) on the host: a gzip file is cut into chunks the
// way host/snk_dgunzip.h does it -- block starts found by probe_header(), every chunk decoded with an unknown window to 16-bit
// symbols, windows chained, markers resolved -- and the bytes are handed back for comparison with zlib (tests/test_inflate_emul.py).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "snk_inflate_core.hip.h"

using namespace snkinf;

// the wave-cooperative I/O of the decoder (Coop: LDS ring for the input, LDS buffer + lane-parallel copies for the output), lanes
// emulated in turn; snk_emul_set_coop(1) selects it for every decode below
static int g_coop = 0;
static uint8_t g_ring[2 * HALF];
static u32 g_qdst[QCAP], g_qinfo[QCAP];
static u16 g_hist[HS];
extern "C" void snk_emul_set_coop(int on) { g_coop = on; }
static Coop *coop() {
    static Coop c;
    if (!g_coop) return nullptr;
    c.ring = g_ring; c.hist = g_hist; c.qdst = g_qdst; c.qinfo = g_qinfo; c.ring_lo = c.ring_end = 0; c.qn = 0; c.q_first = 0;
    return &c;
}

// returns the number of bytes produced, or -(1000 + code) ; info[0] = chunks decoded, info[1] = chunks whose start was found by
// the search, info[2] = markers resolved, info[3] = members seen
extern "C" long snk_emul_gunzip(const uint8_t *gz, size_t n, size_t chunk_bytes, uint8_t *out, size_t out_cap, long *info, long ends_cap) {
    std::vector<uint8_t> comp(n + PAD, 0);
    memcpy(comp.data(), gz, n);
    static Tables T;
    static Scratch S;
    static u32 cl_tab[128];
    // first block of the first member
    Bits hb;
    bits_init(hb, comp.data(), n, 0);
    if (!gzip_header(hb)) return -1001;
    const u64 first_bit = bitpos(hb);
    const size_t nch = (n + chunk_bytes - 1) / chunk_bytes;
    std::vector<Chunk> ck(nch);
    long found = 0;
    for (size_t c = 0; c < nch; ++c) {
        memset(&ck[c], 0, sizeof(Chunk));
        ck[c].start_bit = ~0ull;
        if (c == 0) { ck[c].start_bit = first_bit; ck[c].first_of_member = 1; continue; }
        const u64 lo = std::max<u64>((u64)c * chunk_bytes * 8, first_bit + 1), hi = std::min<u64>((u64)(c + 1) * chunk_bytes * 8, (u64)n * 8);
        for (u64 bit = lo; bit < hi; ++bit)
            if (probe_quick(comp.data(), n, bit) && probe_header(comp.data(), n, bit, T, S, cl_tab)) { ck[c].start_bit = bit; ++found; break; }
    }
    std::vector<u16> syms;
    const u32 cap = (u32)std::min<size_t>(out_cap + 1024, (size_t)1 << 30);
    std::vector<uint8_t> win(WIN, 0);
    size_t produced = 0;
    long decoded = 0, markers = 0, members = 0;
    u64 expect = first_bit;
    for (size_t c = 0; c < nch; ++c) {
        if (ck[c].start_bit == ~0ull) continue;
        if (ck[c].start_bit != expect) {
            if (ck[c].start_bit < expect) continue;       // a found start the chain stepped over: a false positive, skipped
            return -1002;                                 // the chain does not meet
        }
        u64 stop = (u64)n * 8 + 64;
        for (size_t d = c + 1; d < nch; ++d) if (ck[d].start_bit != ~0ull) { stop = ck[d].start_bit; break; }
        ck[c].stop_bit = stop;
        ck[c].out_off = 0;
        ck[c].out_cap = cap;
        syms.assign(cap, 0);
        std::vector<MemberEnd> ends((size_t)ends_cap);
        ck[c].ends_off = 0;
        ck[c].ends_cap = (u32)ends_cap;
        decode_chunk(comp.data(), n, ck[c], syms.data(), ends.data(), T, S, cl_tab, coop());
        if (ck[c].status != INF_OK) return -(1000 + 10 * (long)ck[c].status + 3);
        ++decoded;
        members += ck[c].n_ends;
        if (produced + ck[c].n_syms > out_cap) return -1004;
        for (u32 i = 0; i < ck[c].n_syms; ++i) {
            const u16 s = syms[i];
            if (s >= 256) { ++markers; if (ck[c].known_from != 0xFFFFFFFFu && i >= ck[c].known_from) return -1005; }
            out[produced + i] = s < 256 ? (uint8_t)s : win[s - 256];
        }
        produced += ck[c].n_syms;
        // the next chunk's window: the last 32 KiB of everything so far
        if (produced >= WIN) memcpy(win.data(), out + produced - WIN, WIN);
        else { memset(win.data(), 0, WIN); memcpy(win.data() + WIN - produced, out, produced); }
        expect = ck[c].end_bit;
        if (ck[c].stream_end) break;
    }
    if (info) { info[0] = decoded; info[1] = found; info[2] = markers; info[3] = members; }
    return (long)produced;
}

extern "C" int snk_emul_probe(const uint8_t *gz, size_t n, uint64_t bit) {
    std::vector<uint8_t> comp(n + PAD, 0);
    memcpy(comp.data(), gz, n);
    static Tables T;
    static Scratch S;
    static u32 cl_tab[128];
    const bool full = probe_header(comp.data(), n, bit, T, S, cl_tab), quick = probe_quick(comp.data(), n, bit);
    if (full && !quick) return -1;                      // the quick screen must never reject what the full check accepts
    return (full ? 1 : 0) | (quick ? 2 : 0);
}

// ---- the host orchestration of the device inflate (soapnuke_amd/host/snk_dgunzip.h) over a CPU backend: the two device calls of
// include/snk_gunzip.h done with the same core functions the kernels call (search: quick screen + full check per bit offset;
// decode: decode_chunk per chunk; resolve: chain_byte / resolve_sym per element)
#include "../../soapnuke_amd/host/snk_dgunzip.h"

namespace {
struct EmulBackend : snk::DgBackend {
    uint32_t chunk_bytes, spc, epc;
    std::vector<uint8_t> comp, text, text1;
    std::vector<u16> syms;
    std::vector<Chunk> ck;
    std::string err;
    long decodes = 0, resolves = 0;
    bool decode(const uint8_t *h_comp, uint64_t nbytes, uint64_t first_bit, bool first_of_member, snk_gunzip_chunk *chunks, snk_gunzip_member *ends) override {
        static Tables T;
        static Scratch S;
        static u32 cl_tab[128];
        ++decodes;
        comp.assign(nbytes + PAD, 0);
        memcpy(comp.data(), h_comp, nbytes);
        const uint32_t nc = (uint32_t)((nbytes + chunk_bytes - 1) / chunk_bytes);
        std::vector<u64> start(nc, ~0ull);
        for (uint32_t c = 1; c < nc; ++c) {                                    // inf_search_kernel
            u64 lo = (u64)c * chunk_bytes * 8;
            const u64 hi = std::min<u64>((u64)(c + 1) * chunk_bytes * 8, nbytes * 8);
            if (lo <= first_bit) lo = first_bit + 1;
            for (u64 bit = lo; bit < hi; ++bit)
                if (probe_quick(comp.data(), nbytes, bit) && probe_header(comp.data(), nbytes, bit, T, S, cl_tab)) { start[c] = bit; break; }
        }
        start[0] = first_bit;
        ck.assign(nc, Chunk());
        syms.assign((size_t)nc * spc, 0);
        u64 next = nbytes * 8 + 64;
        for (uint32_t c = nc; c-- > 0;) {                                        // snk_gunzip_decode()
            memset(&ck[c], 0, sizeof(Chunk));
            ck[c].start_bit = start[c];
            ck[c].stop_bit = next;
            ck[c].out_off = (u64)c * spc;
            ck[c].out_cap = spc;
            ck[c].first_of_member = (c == 0 && first_of_member) ? 1u : 0u;
            ck[c].ends_off = c * epc;
            ck[c].ends_cap = epc;
            if (start[c] != ~0ull) next = start[c];
        }
        for (uint32_t c = 0; c < nc; ++c) decode_chunk(comp.data(), nbytes, ck[c], syms.data(), (MemberEnd *)ends, T, S, cl_tab, coop());   // inf_decode_kernel
        memcpy(chunks, ck.data(), nc * sizeof(Chunk));
        return true;
    }
    bool resolve(const uint32_t *order, uint32_t k, const uint8_t *win_in, uint8_t *h_text, uint64_t text_bytes, uint8_t *win_out) override {
        ++resolves;
        std::vector<uint8_t> w(WIN, 0), nw(WIN);
        if (win_in) memcpy(w.data(), win_in, WIN);
        u64 at = 0;
        for (uint32_t j = 0; j < k; ++j) {
            const Chunk &c = ck[order[j]];
            const u16 *s = syms.data() + c.out_off;
            for (u32 i = 0; i < c.n_syms; ++i) h_text[at + i] = resolve_sym(s[i], w.data());     // inf_resolve_kernel
            for (u32 i = 0; i < (u32)WIN; ++i) nw[i] = chain_byte(c.n_syms, s, w.data(), i);     // inf_chain_kernel
            w.swap(nw);
            at += c.n_syms;
        }
        if (at != text_bytes) { err = "text_bytes mismatch"; return false; }
        memcpy(win_out, w.data(), WIN);
        return true;
    }
    uint8_t *text_buffer(int slot, size_t bytes) override { auto &t = slot ? text1 : text; if (t.size() < bytes) t.assign(bytes, 0); return t.data(); }
    std::string error() override { return err; }
};
}  // namespace

// info: windows, fallback bit (-1: none), backend decode calls, resumes << 16 | host spells
extern "C" long snk_emul_dgunzip(const uint8_t *gz, size_t n, size_t window, uint32_t chunk_bytes, uint32_t spc, uint32_t epc, uint8_t *out, size_t out_cap,
                                 long *info, char *errbuf, size_t errcap) {
    EmulBackend be;
    be.chunk_bytes = chunk_bytes; be.spc = spc; be.epc = epc;
    snk::DeviceGunzip::Geometry g{window, chunk_bytes, spc, epc};
    snk::DeviceGunzip z(gz, n, &be, g, 3);
    size_t got = 0;
    while (!z.done() && !z.error()) {
        if (got == out_cap) break;
        got += z.run(out + got, std::min<size_t>(out_cap - got, 777777));
    }
    if (info) { info[0] = (long)z.windows(); info[1] = z.fallback_bit() == ~0ull ? -1 : (long)z.fallback_bit(); info[2] = be.decodes; info[3] = (long)((z.resumes() << 16) | z.host_spells()); }
    if (z.error()) { if (errbuf && errcap) { strncpy(errbuf, z.error(), errcap - 1); errbuf[errcap - 1] = 0; } return -1; }
    return (long)got;
}

extern "C" void snk_emul_stats(unsigned long long *out5, int reset) {
    HostStats &h = host_stats();
    out5[0] = h.literals; out5[1] = h.matches; out5[2] = h.match_syms; out5[3] = h.flush_conflict; out5[4] = h.flush_full;
    for (int k = 0; k < 8; ++k) out5[5 + k] = h.dist_lt[k];
    out5[13] = h.near;
    if (reset) h = HostStats();
}

#ifdef SNK_EMUL_MAIN
// Stand-alone runner for the sanitizer builds (tests/test_sanitizers.py): file.gz file.raw window chunk_bytes syms_per_chunk
//   exit 0 + "IDENTICAL", 2 + "ERROR <text>" for a stream the decoder refuses, 1 for a difference.
#include <fstream>
#include <iterator>
static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s file.gz file.raw window chunk_bytes syms_per_chunk\n", argv[0]); return 64; }
    const std::vector<uint8_t> gz = slurp(argv[1]), want = slurp(argv[2]);
    const size_t window = (size_t)atol(argv[3]);
    const uint32_t chunk = (uint32_t)atol(argv[4]), spc = (uint32_t)atol(argv[5]);
    {   // a reader that walks away after the first bytes: the producer is inside a window or waiting for a slot
        EmulBackend be;
        be.chunk_bytes = chunk; be.spc = spc; be.epc = 16;
        snk::DeviceGunzip z(gz.data(), gz.size(), &be, snk::DeviceGunzip::Geometry{window, chunk, spc, 16}, 3);
        uint8_t few[100];
        (void)z.run(few, sizeof few);
    }
    std::vector<uint8_t> out(want.size() + 64);
    long info[4] = {0, 0, 0, 0};
    char err[256] = "";
    const long r = snk_emul_dgunzip(gz.data(), gz.size(), window, chunk, spc, 16, out.data(), want.size() + 32, info, err, sizeof err);
    if (r < 0) { printf("ERROR %s\n", err); return 2; }
    if ((size_t)r != want.size() || memcmp(out.data(), want.data(), want.size()) != 0) { printf("DIFFERENT (%ld of %zu bytes)\n", r, want.size()); return 1; }
    printf("IDENTICAL windows=%ld fallback=%ld decodes=%ld\n", info[0], info[1], info[2]);
    return 0;
}
#endif
