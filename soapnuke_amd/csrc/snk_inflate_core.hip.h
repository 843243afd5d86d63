// snk_inflate_core.hip.h -- DEFLATE (RFC 1951) decoding of one chunk of a gzip stream with an UNKNOWN 32 KiB window, as plain
// single-thread code that compiles for gfx950 (snk_inflate.hip: one wavefront per chunk, the tables in LDS) and for the host
// (tests/host_emul/: the same functions against zlib's bytes, no GPU needed).
//
// It replaces the reference's reading loop -- gzgets() over one zlib inflate stream per file, src/peprocess.cpp:2063-2113 -- by the
// two-pass scheme of pugz / rapidgzip that host/snk_pgunzip.h already runs on the CPU:
//   * a chunk starts at a dynamic-Huffman block header found by probing bit offsets (probe_header(): the checks zlib itself
//     makes on a header -- code counts, a complete code-length code, lengths that decode without overrun, complete literal /
//     distance codes, an end-of-block code);
//   * it is decoded to 16-bit symbols: 0..255 a byte, 256 + k "byte k of the 32 KiB in front of this chunk" (a MARKER; matches
//     copy markers like literals); a gzip member that ends inside the chunk is followed through its trailer and the next
//     member's header, after which nothing is unknown any more;
//   * later kernels chain the windows and replace the markers.
// Everything zlib rejects is rejected here (status INF_BAD): the caller falls back to the host decoder, so the bytes are always
// zlib's bytes or an error.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SNKI_DEV __device__ __forceinline__
#else
#define SNKI_DEV inline
#endif

namespace snkinf {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

enum { LIT_ROOT = 10, DIST_ROOT = 7, LIT_CAP = 2048, DIST_CAP = 1024, WIN = 32768, PAD = 4096 };   // PAD: zero bytes the caller keeps behind the compressed bytes
enum { INF_OK = 0, INF_FULL = 1, INF_BAD = 2, INF_TOO_MANY_MEMBERS = 3, INF_NOT_STARTED = 4 };
enum { T_INVALID = 0, T_LIT = 1, T_LEN = 2, T_EOB = 3, T_SUB = 4, T_DIST = 5 };

struct MemberEnd { u32 sym_index; u32 crc; u32 isize; u32 pad_; };     // the member's text ends in front of symbol sym_index of the chunk

// one chunk: what the host asks for and what the decoder reports
struct Chunk {
    u64 start_bit;            // in: the block header the chunk starts at (bit offset into the compressed bytes); ~0: none, nothing to do
    u64 stop_bit;             // in: stop in front of the first block header at or past this bit (the next chunk's start)
    u64 out_off;              // in: first symbol slot of this chunk
    u32 out_cap;              // in: symbol slots
    u32 first_of_member;      // in: 1 = the chunk starts with a member's first block (empty window, nothing unknown)
    u32 n_syms;               // out
    u32 status;               // out: INF_*
    u64 end_bit;              // out: where decoding stopped (a block header, or the end of the stream's last member)
    u32 known_from;           // out: symbols from this index on belong to a member that started inside the chunk (no markers); 0xFFFFFFFF: none
    u32 n_ends;               // out: members that ended inside the chunk, reported in ends[ends_off ...]
    u32 stream_end;           // out: 1 = the input ended behind a member trailer (end of the gzip file)
    u32 ends_off, ends_cap;   // in: this chunk's slots of the member-end array (BGZF-style files hold a member per few KB)
    u32 pad_[3];
};

struct Tables { u32 lit[LIT_CAP]; u32 dist[DIST_CAP]; };
struct Scratch { u8 lens[320]; u16 count[16], offs[16]; u16 sorted[288]; u8 submax[1 << LIT_ROOT]; };   // header / table-building workspace

// ---------------------------------------------------------------- wave-cooperative I/O (optional)
// The plain form of the decoder is one thread that loads its input from global memory and stores every symbol there.  On the GPU
// a chunk is decoded by a whole wavefront whose 64 lanes run the same (uniform) code; with a Coop they share the I/O: the compressed
// bytes come through a 2 KB ring in LDS that all lanes refill with 16-byte loads, and matches are copied 64 at a time, one per lane.
// SNKI_LANES(lane): the statement runs for this lane on the device and for the 64 lanes in turn on the host (the emulation) -- in
// DESCENDING order, so that a lane-parallel step whose lanes did depend on each other would show (ascending order is the
// sequential order, in which everything is right).
// Lanes hand data to each other through LDS at the marked places; a wave runs in lock-step and its LDS operations complete in
// order, so there is nothing to emit on the device (snk_device.h has the same definition for the other kernels; tests/simt, which
// runs the lanes one after the other, makes it the place where they wait for each other).
#ifndef SNK_WAVE_SYNC
#if defined(__HIP_DEVICE_COMPILE__)
// compiler-level ordering only: a wavefront-scope fence pair around a wave barrier emits no instruction on gfx950, but it stops
// LLVM from moving one lane's LDS / memory load across another lane's store at this point
#define SNK_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define SNK_WAVE_SYNC() ((void)0)
#endif
#endif
#if defined(__HIPCC__)
#define SNKI_LANES(lane) for (int lane = (int)(threadIdx.x & 63), once_ = 1; once_; once_ = 0)
#define SNKI_FENCE() __threadfence_block()
// The decoder's tables and header workspace are written by all 64 lanes at once with the same values (the lanes run the same
// code on the same data): one copy per wave in LDS.  (tests/simt: a copy per lane -- a lane-after-lane emulation would apply a
// read-modify-write of such a word 64 times.)
#ifndef SNK_WAVE_UNIFORM_SHARED
#define SNK_WAVE_UNIFORM_SHARED __shared__
#endif
#else
#define SNKI_LANES(lane) for (int lane = 63; lane >= 0; --lane)
#define SNKI_FENCE() ((void)0)
#endif
enum { HALF = 1024, QCAP = 64, HS = 4096, NEAR_MAX = HS - 258, SPAN_MAX = HS - 2 * 258 };
struct Coop {
    u8 *ring;                 // LDS, 2 * HALF bytes: compressed bytes [ring_lo, ring_end), byte p at ring[p % (2 * HALF)]
    u64 ring_lo, ring_end;    // multiples of HALF, ring_end - ring_lo <= 2 * HALF
    // gzip turns FASTQ into short matches (tools/inflate_stats.py: 82 matches of 3.8 symbols and 14 literals per read at level 1, 52 of
    // 5.4 and 46 at level 6) whose copies are chains of memory round trips.  The last HS symbols are kept in LDS: a NEAR match
    // (distance <= NEAR_MAX: 71 % at level 1, 37 % at level 6) is copied there at once, lane-parallel.  A FAR match waits in a queue
    // and the queue is copied QCAP matches at a time, one per lane, from global memory -- 64 chains in flight hide the round trips,
    // and decoding does not wait for them (the Huffman stream does not depend on the output).  The queue runs early when a near
    // match wants a symbol a queued match has yet to produce (the lanes compare their match with the new source range), and before
    // it spans SPAN_MAX symbols (a far source then never lies inside it).
    u16 *hist;                // LDS, HS symbols: symbol i at hist[i % HS] (unknown while a queued match owns it)
    u32 *qdst, *qinfo;        // LDS, QCAP words each: destination index; distance | length << 16
    u32 qn, q_first;          // queued matches; the destination of the oldest of them
};

// ---------------------------------------------------------------- bit input
// 8 bytes from any address out of aligned dword loads (the compressed buffer is padded with PAD zero bytes behind its end)
SNKI_DEV u64 load64(const u8 *base, u64 pos) {
    const u32 *w = reinterpret_cast<const u32 *>(base + (pos & ~3ull));
    const u32 sh = (u32)(pos & 3) * 8;
    const u64 lo = ((u64)w[1] << 32) | w[0], hi = w[2];
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

struct Bits {
    const u8 *base;
    u64 nbytes;               // valid compressed bytes (the buffer holds PAD zero bytes more)
    u64 pos;                  // next byte to load
    u64 bb;
    int bc;                   // valid bits in bb
    Coop *co;                 // null: loads straight from global memory
    bool coop;                // co != nullptr, kept as a flag: a null test of the pointer is a comparison of the caller's Coop object's
                              // address, and an object whose address is compared is not promoted to registers (it stayed in scratch memory,
                              // its LDS pointers generic: scratch loads + flat accesses on every symbol)
};
// the same 8 bytes through the ring
SNKI_DEV u64 ring64(Bits &b, u64 pos) {
    Coop &c = *b.co;
    if (pos < c.ring_lo || (pos & ~(u64)(HALF - 1)) > c.ring_end)
        c.ring_lo = c.ring_end = pos & ~(u64)(HALF - 1);                      // a jump (stored block, member header): start over at its half
    while (pos + 12 > c.ring_end) {                                          // (uniform) the next half, 16 bytes per lane
        const u64 from = c.ring_end;
        SNK_WAVE_SYNC();
        SNKI_LANES(lane) {
            const u32 *src = reinterpret_cast<const u32 *>(b.base + from + (u64)lane * 16);
            u32 *dst = reinterpret_cast<u32 *>(c.ring + (from & (u64)HALF) + (u64)lane * 16);
            const u32 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
            dst[0] = w0; dst[1] = w1; dst[2] = w2; dst[3] = w3;
        }
        SNK_WAVE_SYNC();
        c.ring_end = from + HALF;
        if (c.ring_end - c.ring_lo > 2 * HALF) c.ring_lo = c.ring_end - 2 * HALF;
    }
    const u32 a = (u32)(pos & ~3ull), M = 2 * HALF - 1;
    const u32 w0 = *reinterpret_cast<const u32 *>(c.ring + (a & M)), w1 = *reinterpret_cast<const u32 *>(c.ring + ((a + 4) & M)),
              w2 = *reinterpret_cast<const u32 *>(c.ring + ((a + 8) & M));
    const u32 sh = (u32)(pos & 3) * 8;
    const u64 lo = ((u64)w1 << 32) | w0;
    return sh ? (lo >> sh) | ((u64)w2 << (64 - sh)) : lo;
}
SNKI_DEV u64 fetch64(Bits &b, u64 pos) {            // zeros behind the end of the input (the caller notices with past_end())
    const u64 p = pos < b.nbytes + 8 ? pos : b.nbytes + 8;
    return b.coop ? ring64(b, p) : load64(b.base, p);
}
SNKI_DEV void bits_init(Bits &b, const u8 *base, u64 nbytes, u64 bit, Coop *co = nullptr, bool coop = false) {
    b.base = base; b.nbytes = nbytes; b.pos = bit >> 3; b.bb = 0; b.bc = 0; b.co = co; b.coop = coop;
    const int skip = (int)(bit & 7);
    if (skip) { b.bb = fetch64(b, b.pos); b.pos += 7; b.bc = 56; b.bb &= (1ull << 56) - 1; b.bb >>= skip; b.bc -= skip; }
}
SNKI_DEV void refill(Bits &b) {                     // at least 56 valid bits afterwards
    b.bb |= fetch64(b, b.pos) << b.bc;
    b.pos += (u64)((63 - b.bc) >> 3);
    b.bc |= 56;
}
SNKI_DEV u64 bitpos(const Bits &b) { return b.pos * 8 - (u64)b.bc; }
SNKI_DEV u32 take(Bits &b, int n) {                 // n <= 32, n <= bc
    const u32 v = (u32)(b.bb & ((1ull << n) - 1));
    b.bb >>= n; b.bc -= n;
    return v;
}
SNKI_DEV bool past_end(const Bits &b) { return bitpos(b) > b.nbytes * 8; }

// ---------------------------------------------------------------- tables
SNKI_DEV u32 entry(u32 val, u32 extra, u32 type, u32 nbits) { return (val << 16) | (extra << 12) | (type << 8) | nbits; }

SNKI_DEV u32 rev_bits(u32 code, int len) {
    u32 r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// length / distance symbols: base value and extra bits (RFC 1951 3.2.5)
SNKI_DEV void len_sym(int s, u32 &base, u32 &extra) {       // s = symbol - 257, 0..28
    if (s < 8) { base = 3 + s; extra = 0; }
    else if (s == 28) { base = 258; extra = 0; }
    else { extra = (u32)(s - 4) >> 2; base = 3 + ((4 + ((u32)s & 3)) << extra); }
}
SNKI_DEV void dist_sym(int s, u32 &base, u32 &extra) {      // s = 0..29
    if (s < 4) { base = 1 + s; extra = 0; }
    else { extra = (u32)(s - 2) >> 1; base = 1 + ((2 + ((u32)s & 1)) << extra); }
}

// Canonical Huffman decoding table from code lengths (two levels: `root` index bits, then one sub-table per long prefix).
// kind: 0 = literal/length alphabet, 1 = distances.  False: over-subscribed, incomplete (unless it is a single one-bit code, which
// zlib lets pass), no end-of-block code, or the table would not fit.
SNKI_DEV bool build_table(u32 *tab, int cap, int root, const u8 *lens, int n, int kind, Scratch &S) {
    for (int i = 0; i < 16; ++i) S.count[i] = 0;
    for (int s = 0; s < n; ++s) S.count[lens[s]]++;
    int maxl = 15;
    while (maxl > 0 && S.count[maxl] == 0) --maxl;
    const int size = 1 << root;
    if (maxl == 0) {                                   // no codes at all: every look-up is an error (zlib builds the same for distances)
        if (kind == 0) return false;
        for (int i = 0; i < size; ++i) tab[i] = entry(0, 0, T_INVALID, 1);
        return true;
    }
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - (int)S.count[l]; if (left < 0) return false; }
    if (left > 0 && maxl != 1) return false;           // incomplete set
    if (kind == 0 && lens[256] == 0) return false;     // missing end-of-block
    // symbols in canonical order
    S.offs[1] = 0;
    for (int l = 1; l < 15; ++l) S.offs[l + 1] = (u16)(S.offs[l] + S.count[l]);
    for (int s = 0; s < n; ++s) if (lens[s]) S.sorted[S.offs[lens[s]]++] = (u16)s;
    for (int i = 0; i < size; ++i) { tab[i] = entry(0, 0, T_INVALID, 1); S.submax[i] = 0; }
    // first pass over the long codes: the longest code behind every root prefix
    const int total = (int)S.offs[15];
    u32 code = 0;
    int k = 0;
    for (int l = 1; l <= maxl; ++l) {
        for (int c = 0; c < (int)S.count[l]; ++c, ++k, ++code) {
            if (l > root) { const u32 p = rev_bits(code, l) & (u32)(size - 1); if (S.submax[p] < l) S.submax[p] = (u8)l; }
        }
        code <<= 1;
    }
    int next_sub = size;
    code = 0;
    k = 0;
    for (int l = 1; l <= maxl; ++l) {
        for (int c = 0; c < (int)S.count[l]; ++c, ++k, ++code) {
            const int s = (int)S.sorted[k];
            u32 e;
            if (kind == 0) {
                if (s < 256) e = entry((u32)s, 0, T_LIT, 0);
                else if (s == 256) e = entry(0, 0, T_EOB, 0);
                else if (s <= 285) { u32 b, x; len_sym(s - 257, b, x); e = entry(b, x, T_LEN, 0); }
                else e = entry(0, 0, T_INVALID, 0);          // 286, 287: in the fixed code, never valid
            } else {
                if (s <= 29) { u32 b, x; dist_sym(s, b, x); e = entry(b, x, T_DIST, 0); }
                else e = entry(0, 0, T_INVALID, 0);
            }
            const u32 r = rev_bits(code, l);
            if (l <= root) {
                e |= (u32)l;
                for (u32 i = r; i < (u32)size; i += 1u << l) tab[i] = e;
            } else {
                const u32 p = r & (u32)(size - 1);
                const int sb = (int)S.submax[p] - root;                      // index bits of this prefix's sub-table
                if (((tab[p] >> 8) & 15) != T_SUB) {
                    if (next_sub + (1 << sb) > cap) return false;
                    tab[p] = entry((u32)next_sub, (u32)sb, T_SUB, (u32)root);
                    for (int i = 0; i < (1 << sb); ++i) tab[next_sub + i] = entry(0, 0, T_INVALID, 1);
                    next_sub += 1 << sb;
                }
                const u32 sub = tab[p] >> 16;
                e |= (u32)(l - root);
                for (u32 i = r >> root; i < (1u << sb); i += 1u << (l - root)) tab[sub + i] = e;
            }
        }
        code <<= 1;
    }
    (void)total;
    return true;
}

SNKI_DEV u32 lookup(const u32 *tab, int root, Bits &b) {    // consumes the code's bits; the entry of its symbol
    u32 e = tab[b.bb & ((1u << root) - 1)];
    if (((e >> 8) & 15) == T_SUB) {
        b.bb >>= root; b.bc -= root;
        e = tab[(e >> 16) + (u32)(b.bb & ((1u << ((e >> 12) & 15)) - 1))];
    }
    const int n = (int)(e & 0xFF);
    b.bb >>= n; b.bc -= n;
    return e;
}

// ---------------------------------------------------------------- block headers
// the code lengths of a dynamic block (RFC 1951 3.2.7) into S.lens[0 .. nlit + ndist); false: what zlib calls an invalid header
SNKI_DEV bool dynamic_lengths(Bits &b, Scratch &S, u32 *cl_tab /* 128 entries */, int &nlit, int &ndist) {
    refill(b);
    nlit = (int)take(b, 5) + 257;
    ndist = (int)take(b, 5) + 1;
    const int ncl = (int)take(b, 4) + 4;
    if (nlit > 286 || ndist > 30) return false;
    u8 cl[19];
    for (int i = 0; i < 19; ++i) cl[i] = 0;
    const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    refill(b);                                       // 42 bits of header taken so far at most 14; 19 x 3 = 57 more: two refills
    for (int i = 0; i < ncl; ++i) {
        if (b.bc < 3) refill(b);
        cl[order[i]] = (u8)take(b, 3);
    }
    // the code-length code: at most 7 bits -> a flat 128-entry table; it must be complete (zlib: type CODES)
    {
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 19; ++i) cnt[cl[i]]++;
        int left = 1, any = 0;
        for (int l = 1; l <= 7; ++l) { left = (left << 1) - cnt[l]; any += cnt[l]; if (left < 0) return false; }
        if (any == 0 || left > 0) return false;
        u32 code = 0;
        for (int l = 1; l <= 7; ++l) {
            for (int s = 0; s < 19; ++s) {
                if (cl[s] != l) continue;
                const u32 r = rev_bits(code, l);
                for (u32 i = r; i < 128; i += 1u << l) cl_tab[i] = ((u32)s << 8) | (u32)l;
                ++code;
            }
            code <<= 1;
        }
    }
    int have = 0;
    const int want = nlit + ndist;
    while (have < want) {
        refill(b);
        const u32 e = cl_tab[b.bb & 127];
        const int l = (int)(e & 0xFF), s = (int)(e >> 8);
        take(b, l);
        if (s < 16) { S.lens[have++] = (u8)s; continue; }
        int rep, v = 0;
        if (s == 16) { if (have == 0) return false; v = S.lens[have - 1]; rep = 3 + (int)take(b, 2); }
        else if (s == 17) rep = 3 + (int)take(b, 3);
        else rep = 11 + (int)take(b, 7);
        if (have + rep > want) return false;
        for (int i = 0; i < rep; ++i) S.lens[have++] = (u8)v;
    }
    return !past_end(b);
}

SNKI_DEV void fixed_lengths(Scratch &S) {
    for (int i = 0; i < 144; ++i) S.lens[i] = 8;
    for (int i = 144; i < 256; ++i) S.lens[i] = 9;
    for (int i = 256; i < 280; ++i) S.lens[i] = 7;
    for (int i = 280; i < 288; ++i) S.lens[i] = 8;
    for (int i = 0; i < 32; ++i) S.lens[288 + i] = 5;
}

// The cheap part of probe_header(), registers only (the search runs it in every lane on 64 bit offsets at once): block type,
// code counts, and a complete code-length code.  About one offset in a thousand passes.
SNKI_DEV bool probe_quick(const u8 *comp, u64 nbytes, u64 bit) {
    if (bit + 64 > nbytes * 8) return false;
    u64 w = load64(comp, bit >> 3) >> (bit & 7);                        // 57+ valid bits
    if ((w & 7) != 4) return false;
    w >>= 3;
    const u32 hl = (u32)(w & 31), hd = (u32)((w >> 5) & 31), ncl = (u32)((w >> 10) & 15) + 4;
    if (hl > 29 || hd > 29) return false;
    // the code-length code's lengths: ncl x 3 bits from bit + 17
    const u64 b2 = bit + 17;
    u64 x = load64(comp, b2 >> 3) >> (b2 & 7);                          // 57 valid bits = 19 lengths
    u32 left = 1u << 7, any = 0;                                        // Kraft sum in units of 2^-7
    for (u32 i = 0; i < ncl; ++i) {
        const u32 l = (u32)(x & 7);
        x >>= 3;
        if (l) { const u32 c = 128u >> l; if (c > left) return false; left -= c; any = 1; }
    }
    return any && left == 0;
}

// Is there a non-final dynamic-Huffman block header at this bit?  (the search for a chunk's first block: every check zlib makes
// on such a header.)  cl_tab: 128 words of scratch, T: table space (destroyed).
SNKI_DEV bool probe_header(const u8 *comp, u64 nbytes, u64 bit, Tables &T, Scratch &S, u32 *cl_tab) {
    if (bit + 64 > nbytes * 8) return false;
    Bits b;
    bits_init(b, comp, nbytes, bit);
    refill(b);
    if ((b.bb & 7) != 4) return false;              // BFINAL = 0, BTYPE = 10b (bits: final, then type LSB first)
    take(b, 3);
    {   // cheap rejections first: code counts in range
        const u32 hl = (u32)(b.bb & 31), hd = (u32)((b.bb >> 5) & 31);
        if (hl > 29 || hd > 29) return false;
    }
    int nlit, ndist;
    if (!dynamic_lengths(b, S, cl_tab, nlit, ndist)) return false;
    if (!build_table(T.lit, LIT_CAP, LIT_ROOT, S.lens, nlit, 0, S)) return false;
    return build_table(T.dist, DIST_CAP, DIST_ROOT, S.lens + nlit, ndist, 1, S);
}

// ---------------------------------------------------------------- gzip framing (RFC 1952)
// member header at the (byte-aligned) position of b; false: not a gzip header / truncated
SNKI_DEV bool gzip_header(Bits &b) {
    // work on bytes: drop the bit buffer
    u64 p = (bitpos(b) + 7) >> 3;
    const u8 *d = b.base;
    const u64 n = b.nbytes;
    if (p + 10 > n) return false;
    if (d[p] != 0x1F || d[p + 1] != 0x8B || d[p + 2] != 8) return false;
    const u32 flg = d[p + 3];
    if (flg & 0xE0) return false;
    p += 10;
    if (flg & 4) { if (p + 2 > n) return false; const u32 xl = (u32)d[p] | ((u32)d[p + 1] << 8); p += 2 + xl; }
    if (flg & 8) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 16) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 2) p += 2;
    if (p > n) return false;
    bits_init(b, d, n, p * 8, b.co, b.coop);
    return true;
}

// ---------------------------------------------------------------- the chunk
// symbol output: one thread stores and copies in order; a Coop copies near matches in LDS and queues the far ones (see there)
struct Out {
    u16 *out;                 // the chunk's symbol slots
    u32 n;                    // symbols so far (queued matches included)
    Coop *co;
    bool coop;                // co != nullptr (see Bits)
};
SNKI_DEV u16 marker_of(long src) { return (u16)(256 + WIN + src); }          // src < 0: in front of the chunk
SNKI_DEV u16 match_sym(const u16 *out, long src) { return src < 0 ? marker_of(src) : out[src]; }
#if !defined(__HIPCC__)
struct HostStats { unsigned long long literals, matches, match_syms, flush_conflict, flush_full, dist_lt[8], near; };   // dist_lt[k]: matches nearer than 256 << k
inline HostStats &host_stats() { static HostStats h; return h; }
#define SNKI_STAT(x) (host_stats().x)
#endif
SNKI_DEV void out_flush(Out &o) {                      // runs the queued (far) matches, one per lane
    if (!o.coop || o.co->qn == 0) return;
    Coop &c = *o.co;
    SNKI_FENCE();                                      // the literals and earlier copies the sources are (other lanes' stores)
    SNKI_LANES(lane) {
        if ((u32)lane < c.qn) {
            const u32 dst = c.qdst[lane], dist = c.qinfo[lane] & 0xFFFFu, len = c.qinfo[lane] >> 16;
            const long first = (long)dst - (long)dist;
            for (u32 i = 0; i < len; ++i) {
                const u16 v = match_sym(o.out, first + (long)(i % dist));
                o.out[dst + i] = v;
                c.hist[(dst + i) & (HS - 1)] = v;      // (the queue spans less than HS symbols: the slot is still this symbol's)
            }
        }
    }
    SNKI_FENCE();
    c.qn = 0;
}
// does a queued match have to produce a symbol of [a, b) ?
SNKI_DEV bool queue_owns(const Coop &c, long a, long b) {
    bool hit = false;
#if defined(__HIPCC__)
    const u32 lane = threadIdx.x & 63;
    bool mine = false;
    SNK_WAVE_SYNC();                                   // (lane 0 wrote the queue)
    if (lane < c.qn) { const long d = (long)c.qdst[lane], e = d + (long)(c.qinfo[lane] >> 16); mine = d < b && a < e; }
    hit = __any(mine);
#else
    for (u32 k = 0; k < c.qn; ++k) { const long d = (long)c.qdst[k], e = d + (long)(c.qinfo[k] >> 16); hit = hit || (d < b && a < e); }
#endif
    return hit;
}
SNKI_DEV void out_put(Out &o, u16 v) {
#if !defined(__HIPCC__)
    ++SNKI_STAT(literals);
#endif
    if (o.coop) {
        SNKI_LANES(lane) { if (lane == 0) { o.out[o.n] = v; o.co->hist[o.n & (HS - 1)] = v; } }
        ++o.n;
        if (o.co->qn && o.n - o.co->q_first > (u32)SPAN_MAX) out_flush(o);     // (the queue never spans more than the history holds)
    } else {
        o.out[o.n++] = v;
    }
}
// a match of len symbols at distance dist (1..32768)
SNKI_DEV void out_match(Out &o, u32 len, u32 dist) {
    const long first = (long)o.n - (long)dist;
    if (o.coop) {
        Coop &c = *o.co;
#if !defined(__HIPCC__)
        ++SNKI_STAT(matches); SNKI_STAT(match_syms) += len;
        for (int k = 0; k < 8; ++k) if (dist < (256u << k)) ++SNKI_STAT(dist_lt[k]);
#endif
        if (dist <= (u32)NEAR_MAX) {
            // near: from the LDS history, at once -- unless a queued match still owes one of the source symbols
            const u32 used = len < dist ? len : dist;                          // (an overlapping match repeats its first dist symbols)
            if (c.qn && first + (long)used > (long)c.q_first && queue_owns(c, first, first + (long)used)) {
#if !defined(__HIPCC__)
                ++SNKI_STAT(flush_conflict);
#endif
                out_flush(o);
            }
#if !defined(__HIPCC__)
            ++SNKI_STAT(near);
#endif
            SNK_WAVE_SYNC();                           // (the sources: literals stored by lane 0, copies by any lane)
            SNKI_LANES(lane) {
                for (u32 i = (u32)lane; i < len; i += 64) {
                    const long src = first + (long)(i % dist);
                    const u16 v = src < 0 ? marker_of(src) : c.hist[(u32)src & (HS - 1)];
                    o.out[o.n + i] = v;
                    c.hist[(o.n + i) & (HS - 1)] = v;                          // (dist <= HS - len: no source slot is a destination slot)
                }
            }
            o.n += len;
        } else {
            // far: queued; its source lies in front of everything queued (distance > NEAR_MAX, span <= SPAN_MAX) -- checked all the same
            if (c.qn && first + (long)(len < dist ? len : dist) > (long)c.q_first) out_flush(o);
            if (c.qn == 0) c.q_first = o.n;
            SNKI_LANES(lane) { if (lane == 0) { c.qdst[c.qn] = o.n; c.qinfo[c.qn] = dist | (len << 16); } }
            ++c.qn;
            o.n += len;
            if (c.qn == QCAP || o.n - c.q_first > (u32)SPAN_MAX) {
#if !defined(__HIPCC__)
                ++SNKI_STAT(flush_full);
#endif
                out_flush(o);
            }
        }
        if (c.qn && o.n - c.q_first > (u32)SPAN_MAX) out_flush(o);             // (literals and near matches lengthen the span too)
    } else {
        for (u32 i = 0; i < len; ++i) o.out[o.n + i] = match_sym(o.out, first + (long)i);
        o.n += len;
    }
}
SNKI_DEV void out_stored(Out &o, const u8 *comp, u64 p, u32 len) {
    if (o.coop) {
        out_flush(o);
        SNKI_LANES(lane) {
            for (u32 i = (u32)lane; i < len; i += 64) {
                o.out[o.n + i] = comp[p + i];
                if (i + HS >= len) o.co->hist[(o.n + i) & (HS - 1)] = comp[p + i];      // (the last HS of them)
            }
        }
        o.n += len;
    } else {
        for (u32 i = 0; i < len; ++i) o.out[o.n + i] = comp[p + i];
        o.n += len;
    }
}

// Decodes chunk ck (see its fields) to syms[ck.out_off ...].  One thread -- or, with a Coop, the 64 lanes of a wavefront running
// it in lockstep on the same values; T and S are the workspace (LDS on the device).
// (coop says whether there is a Coop: the kernels pass the flag themselves, so that the address of their Coop object is never compared)
SNKI_DEV void decode_chunk(const u8 *comp, u64 nbytes, Chunk &ck, u16 *syms_all, MemberEnd *ends_all, Tables &T, Scratch &S, u32 *cl_tab,
                           Coop *co, bool coop) {
    ck.n_syms = 0; ck.status = INF_OK; ck.end_bit = ck.start_bit; ck.known_from = 0xFFFFFFFFu; ck.n_ends = 0; ck.stream_end = 0;
    if (ck.start_bit == ~0ull) { ck.status = INF_NOT_STARTED; return; }
    Out o;
    o.out = syms_all + ck.out_off; o.n = 0; o.co = co; o.coop = coop;
    if (coop) { co->qn = 0; co->q_first = 0; co->ring_lo = co->ring_end = 0; }
    const u32 cap = ck.out_cap;
    u32 known_from = ck.first_of_member ? 0u : 0xFFFFFFFFu;
    Bits b;
    bits_init(b, comp, nbytes, ck.start_bit, co, coop);
    for (;;) {
        // ---- block header
        const u64 at = bitpos(b);
        ck.end_bit = at;
        if (at >= ck.stop_bit) break;
        refill(b);
        const u32 final_ = take(b, 1), type = take(b, 2);
        if (type == 3) { ck.status = INF_BAD; break; }
        if (type == 0) {                               // stored
            take(b, b.bc & 7);                          // to the byte boundary (bc is a multiple of 8 afterwards)
            refill(b);
            const u32 len = take(b, 16), nlen = take(b, 16);
            if ((len ^ 0xFFFFu) != nlen) { ck.status = INF_BAD; break; }
            if (o.n + len > cap) { ck.status = INF_FULL; break; }
            const u64 p = bitpos(b) >> 3;
            if (p + len > nbytes) { ck.status = INF_BAD; break; }
            out_stored(o, comp, p, len);
            bits_init(b, comp, nbytes, (p + len) * 8, co, coop);
        } else {
            int nlit = 288, ndist = 32;
            if (type == 1) fixed_lengths(S);
            else if (!dynamic_lengths(b, S, cl_tab, nlit, ndist)) { ck.status = INF_BAD; break; }
            if (!build_table(T.lit, LIT_CAP, LIT_ROOT, S.lens, nlit, 0, S) ||
                !build_table(T.dist, DIST_CAP, DIST_ROOT, S.lens + nlit, ndist, 1, S)) { ck.status = INF_BAD; break; }
            // ---- symbols
            bool bad = false, full = false;
            for (;;) {
                if (o.n + 260 > cap) { full = true; break; }
                refill(b);
                u32 e = lookup(T.lit, LIT_ROOT, b);
                u32 t = (e >> 8) & 15;
                if (t == T_LIT) {
                    out_put(o, (u16)(e >> 16));
                    // a second literal from the same refill (56 bits cover two 15-bit codes and a length's extras)
                    e = lookup(T.lit, LIT_ROOT, b);
                    t = (e >> 8) & 15;
                    if (t == T_LIT) { out_put(o, (u16)(e >> 16)); continue; }
                }
                if (t == T_EOB) break;
                if (t != T_LEN) { bad = true; break; }
                const u32 len = (e >> 16) + take(b, (int)((e >> 12) & 15));
                if (b.bc < 32) refill(b);
                const u32 de = lookup(T.dist, DIST_ROOT, b);
                if (((de >> 8) & 15) != T_DIST) { bad = true; break; }
                const u32 dist = (de >> 16) + take(b, (int)((de >> 12) & 15));
                if (known_from != 0xFFFFFFFFu) {
                    if (dist > o.n - known_from) { bad = true; break; }        // reaches in front of its member: zlib's "too far back"
                } else if (dist > o.n && dist > (u32)WIN) { bad = true; break; }
                out_match(o, len, dist);
            }
            if (b.bc < 0 || past_end(b)) bad = true;
            if (bad) { ck.status = INF_BAD; break; }
            if (full) { ck.status = INF_FULL; break; }
        }
        if (final_) {
            // ---- member trailer, then the next member or the end of the file
            take(b, b.bc & 7);
            const u64 p = bitpos(b) >> 3;
            if (p + 8 > nbytes) { ck.status = INF_BAD; break; }
            if (ck.n_ends == ck.ends_cap) { ck.status = INF_TOO_MANY_MEMBERS; break; }
            MemberEnd &m = ends_all[ck.ends_off + ck.n_ends++];
            m.sym_index = o.n;
            m.crc = (u32)comp[p] | ((u32)comp[p + 1] << 8) | ((u32)comp[p + 2] << 16) | ((u32)comp[p + 3] << 24);
            m.isize = (u32)comp[p + 4] | ((u32)comp[p + 5] << 8) | ((u32)comp[p + 6] << 16) | ((u32)comp[p + 7] << 24);
            bits_init(b, comp, nbytes, (p + 8) * 8, co, coop);
            ck.end_bit = (p + 8) * 8;
            if (p + 8 >= nbytes) { ck.stream_end = 1; break; }
            if (!gzip_header(b)) { ck.status = INF_BAD; break; }
            known_from = o.n;
            ck.known_from = o.n;
        }
    }
    out_flush(o);
    ck.n_syms = o.n;
}

// ---------------------------------------------------------------- windows and markers
SNKI_DEV u8 resolve_sym(u16 x, const u8 *win) { return x < 256 ? (u8)x : win[x - 256]; }
// byte i of the window BEHIND a chunk (the last 32 KiB of: the window in front of it ++ its text); s = the chunk's symbols
SNKI_DEV u8 chain_byte(u32 n_syms, const u16 *s, const u8 *win, u32 i) {
    const u32 back = (u32)WIN - i;                     // this many bytes from the end of the stream so far
    return back <= n_syms ? resolve_sym(s[n_syms - back], win) : win[i + n_syms];
}

// for callers that hold a Coop by pointer anyway (the CPU twins of the kernels in tests/host_emul)
SNKI_DEV void decode_chunk(const u8 *comp, u64 nbytes, Chunk &ck, u16 *syms_all, MemberEnd *ends_all, Tables &T, Scratch &S, u32 *cl_tab,
                           Coop *co = nullptr) {
    decode_chunk(comp, nbytes, ck, syms_all, ends_all, T, S, cl_tab, co, co != nullptr);
}

}  // namespace snkinf
