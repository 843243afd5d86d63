// snk_deflate.h -- a fast gzip (RFC 1952 / 1951) ENCODER for the clean-FASTQ writer threads.
//
// The reference compresses its output with zlib level 2 (gzsetparams(..., 2, Z_DEFAULT_STRATEGY), src/peprocess.cpp:1809)
// at ~100 MB/s per thread; with .gz output that is a large part of this CLI's host CPU time, on a host
// whose core count is the budget.  Compressed BYTES are not part of the parity contract ("zlib ... affects only
// compressed bytes"): any valid gzip stream of the same text is equivalent.  Two front ends feed one dynamic-Huffman back
// end (one block per ~512 KiB of input with exact symbol counts, length-limited codes, a 64-bit bit writer that emits five
// literals per drain, CRC-32 by PCLMULQDQ folding -- snk_crc32.h -- and ISIZE per member):
//   deflate_fastq()  the text is FASTQ records.  No search at all: the name and '+' lines are compared with the record
//                    before at the same place (runs of >= 4 equal bytes become matches), bases and qualities go out as
//                    literal runs.  ~4 x the speed of the hash search and a few per cent smaller on FASTQ.
//   deflate_all()    anything else: greedy LZ77 with a single-probe hash table of 4-byte sequences (32 KiB window, matches
//                    of 4..258 bytes extended 8 bytes at a time, striding over stretches without matches); the ratio of
//                    zlib's low levels at about twice their speed.
//
// tests/test_deflate.py: output decompressed by zlib and by snk_inflate.h equals the input for FASTQ of every shape, text
// that only looks like FASTQ, runs, random bytes, empty input, sizes around every block boundary, both front ends.
#ifndef SNK_DEFLATE_H
#define SNK_DEFLATE_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <zlib.h>
#include "snk_crc32.h"
#include <algorithm>
#include <string>
#include <vector>

#ifndef SNK_DEFLATE_MISS_SHIFT
#define SNK_DEFLATE_MISS_SHIFT 4      // the scan strides by 1 + misses / 2^shift
#endif

namespace snk {

class FastDeflate {
public:
    FastDeflate() : head_(1u << HASH_BITS, 0), syms_(BLOCK_SYMS + 64) { memset(lf_, 0, sizeof lf_); memset(df_, 0, sizeof df_); memset(h4_, 0, sizeof h4_); }

    // appends one complete gzip member holding in[0, n) to `out`
    // fastq: the text is FASTQ records (4 lines each) -- record-aware matching instead of the hash search, see deflate_fastq()
    void gzip_member(const uint8_t *in, size_t n, std::string &out, bool fastq = false) {
        const size_t at = out.size();
        // worst case: literals only with 9-bit codes + block headers; stored-size bound with margin
        out.resize(at + n + n / 4 + (n / 65536 + 2) * 600 + 128);
        uint8_t *p = reinterpret_cast<uint8_t *>(&out[at]);
        static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 3};     // deflate, no flags, mtime 0, XFL 4 (fastest), OS 3
        memcpy(p, hdr, 10);
        op_ = p + 10;
        bitbuf_ = 0;
        bitcnt_ = 0;
        if (fastq && n > 0 && in[0] == '@') deflate_fastq(in, n);
        else deflate_all(in, n);
        flush_bits();
        const uint32_t crc = snk::crc32_fast(0, in, n), isz = (uint32_t)n;
        for (int i = 0; i < 4; ++i) *op_++ = (uint8_t)(crc >> (8 * i));
        for (int i = 0; i < 4; ++i) *op_++ = (uint8_t)(isz >> (8 * i));
        out.resize(at + (size_t)(op_ - p));
    }

private:
    enum { HASH_BITS = 15, WINDOW = 32768, MIN_MATCH = 4, MAX_MATCH = 258, BLOCK_SYMS = 1 << 16, BLOCK_BYTES = 1 << 19, LIT_FLUSH = 1 << 14, NLIT = 286, NDIST = 30 };
    uint32_t lf_[NLIT], df_[NDIST];                      // symbol counts of the block being collected
    uint32_t h4_[4][256];                                // ... its literals, four ways
    uint32_t epoch_ = 0;                                 // added to the positions kept in head_: entries of earlier members fall out of range
    struct Sym { uint16_t litlen, dist; };               // dist == 0: a run of `litlen` literal bytes (taken from the input); else a match of that length
    std::vector<uint32_t> head_;                         // hash -> position + 1 of its newest occurrence (0: none)
    std::vector<Sym> syms_;
    // one run of equal bytes in a very long name / '+' line appends its literal run symbols plus l / 258 match pieces behind a
    // single capacity check: the buffer grows instead of trusting the slack (heap write past BLOCK_SYMS + 64)
    inline void push_sym(int &nsym, const Sym &v) {
        if ((size_t)nsym >= syms_.size()) syms_.resize(syms_.size() * 2);
        syms_[(size_t)nsym++] = v;
    }
    uint8_t *op_ = nullptr;
    uint64_t bitbuf_ = 0;
    int bitcnt_ = 0;

    static inline uint32_t load32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
    static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
    static inline uint32_t hash4(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - HASH_BITS); }

    // ---- bit output (LSB first); at most 57 bits pending
    inline void put(uint32_t code, int len) {
        bitbuf_ |= (uint64_t)code << bitcnt_;
        bitcnt_ += len;
    }
    inline void drain() {                                // keeps fewer than 8 bits pending
        memcpy(op_, &bitbuf_, 8);
        const int bytes = bitcnt_ >> 3;
        op_ += bytes;
        bitbuf_ >>= 8 * bytes;
        bitcnt_ &= 7;
    }
    void flush_bits() {
        drain();
        if (bitcnt_) { *op_++ = (uint8_t)bitbuf_; bitbuf_ = 0; bitcnt_ = 0; }
    }

    // ---- static tables
    struct Tables {
        uint8_t len_code[259];       // match length -> length symbol - 257
        uint8_t dist_code_lo[512];   // dist - 1 < 512 -> distance symbol
        uint8_t dist_code_hi[256];   // (dist - 1) >> 7 -> distance symbol, for dist - 1 >= 512
        uint16_t len_base[29], dist_base[30];
        uint8_t len_extra[29], dist_extra[30];
        Tables() {
            static const uint16_t lb[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
            static const uint8_t le[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
            static const uint16_t db[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
            static const uint8_t de[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
            for (int i = 0; i < 29; ++i) { len_base[i] = lb[i]; len_extra[i] = le[i]; }
            for (int i = 0; i < 30; ++i) { dist_base[i] = db[i]; dist_extra[i] = de[i]; }
            for (int l = 3; l <= 258; ++l) {
                int c = 28;
                while (lb[c] > l) --c;
                len_code[l] = (uint8_t)c;
            }
            for (int d = 1; d <= 32768; ++d) {
                int c = 29;
                while (db[c] > d) --c;
                if (d - 1 < 512) dist_code_lo[d - 1] = (uint8_t)c;
                else dist_code_hi[(d - 1) >> 7] = (uint8_t)c;
            }
        }
    };
    static const Tables &tab() { static const Tables t; return t; }
    static inline int dist_sym(uint32_t dist) {
        const Tables &t = tab();
        return dist - 1 < 512 ? t.dist_code_lo[dist - 1] : t.dist_code_hi[(dist - 1) >> 7];
    }

    // ---- Huffman code lengths (limit `maxbits`) and canonical codes (bit-reversed for LSB-first output)
    static void code_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *lens) {
        struct Node { uint64_t w; int a, b; };
        int order[NLIT + 2], m = 0;
        for (int i = 0; i < n; ++i) { lens[i] = 0; if (freq[i]) order[m++] = i; }
        if (m == 0) return;
        if (m == 1) {                                      // a lone symbol: give it a sibling, so that the code is complete
            lens[order[0]] = 1;
            lens[order[0] == 0 ? 1 : 0] = 1;
            return;
        }
        std::sort(order, order + m, [&](int x, int y) { return freq[x] != freq[y] ? freq[x] < freq[y] : x < y; });
        // two-queue Huffman construction: leaves in `order`, internal nodes appended in creation (= weight) order
        Node nodes[2 * (NLIT + 2)];
        for (int i = 0; i < m; ++i) nodes[i] = Node{freq[order[i]], -1, -1};
        int leaf = 0, inner = m, next = m;
        auto take = [&]() -> int {
            if (leaf < m && (inner >= next || nodes[leaf].w <= nodes[inner].w)) return leaf++;
            return inner++;
        };
        while ((m - leaf) + (next - inner) > 1) {
            const int x = take(), y = take();
            nodes[next] = Node{nodes[x].w + nodes[y].w, x, y};
            ++next;
        }
        // depths: the root is the last node; children come before their parents
        int depth[2 * (NLIT + 2)];
        depth[next - 1] = 0;
        for (int i = next - 1; i >= m; --i) { depth[nodes[i].a] = depth[i] + 1; depth[nodes[i].b] = depth[i] + 1; }
        int count[64] = {0};
        bool over = false;
        for (int i = 0; i < m; ++i) {
            int d = depth[i];
            if (d > maxbits) { d = maxbits; over = true; }
            count[d]++;
        }
        if (over) {
            // the clamped lengths over-subscribe the code space.  Classic repair on the per-length counts: drop one code
            // of the longest length, split the deepest shorter code into two one level down -- the Kraft sum falls by
            // one unit of 2^-maxbits per pass and the code stays complete when it reaches one
            unsigned long long total = 0;
            for (int l = maxbits; l > 0; --l) total += (unsigned long long)count[l] << (maxbits - l);
            while (total != (1ull << maxbits)) {
                count[maxbits]--;
                for (int l = maxbits - 1; l > 0; --l)
                    if (count[l]) { count[l]--; count[l + 1] += 2; break; }
                --total;
            }
        }
        // hand the lengths out: the longest codes to the rarest symbols (order[] is ascending by frequency)
        int k = 0;
        for (int bits = maxbits; bits >= 1; --bits)
            for (int c = count[bits]; c > 0; --c) lens[order[k++]] = (uint8_t)bits;
    }
    static void make_codes(const uint8_t *lens, int n, uint16_t *codes) {
        int count[16] = {0};
        for (int i = 0; i < n; ++i) count[lens[i]]++;
        count[0] = 0;
        uint32_t next[16], code = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
        for (int i = 0; i < n; ++i) {
            const int l = lens[i];
            if (!l) { codes[i] = 0; continue; }
            uint32_t c = next[l]++, r = 0;
            for (int k = 0; k < l; ++k) { r = (r << 1) | (c & 1); c >>= 1; }
            codes[i] = (uint16_t)r;
        }
    }

    // ---- one dynamic block: nsym symbols covering the input from `src` on
    void write_block(const uint8_t *src, int nsym, bool final) {
        const Tables &t = tab();
        uint32_t *lf = lf_, *df = df_;
        for (int c = 0; c < 256; ++c) { lf[c] += h4_[0][c] + h4_[1][c] + h4_[2][c] + h4_[3][c]; }
        memset(h4_, 0, sizeof h4_);
        lf[256] = 1;
        uint8_t ll[NLIT], dl[NDIST];
        code_lengths(lf, NLIT, 15, ll);
        code_lengths(df, NDIST, 15, dl);
        memset(lf_, 0, sizeof lf_);
        memset(df_, 0, sizeof df_);
        int nlit = NLIT, ndist = NDIST;
        while (nlit > 257 && ll[nlit - 1] == 0) --nlit;
        while (ndist > 1 && dl[ndist - 1] == 0) --ndist;
        if (ndist == 1 && dl[0] == 0) dl[0] = 1;           // no distance used at all: one dummy code keeps every decoder happy
        uint16_t lc[NLIT], dc[NDIST];
        make_codes(ll, nlit, lc);
        make_codes(dl, ndist, dc);
        // code length sequence with run-length symbols 16 / 17 / 18
        uint8_t seq[NLIT + NDIST], cl_sym[NLIT + NDIST], cl_ext[NLIT + NDIST];
        const int total = nlit + ndist;
        memcpy(seq, ll, nlit);
        memcpy(seq + nlit, dl, ndist);
        int ncl = 0;
        uint32_t cf[19] = {0};
        for (int i = 0; i < total;) {
            int run = 1;
            while (i + run < total && seq[i + run] == seq[i]) ++run;
            const int v = seq[i];
            int left = run;
            if (v == 0) {
                while (left >= 11) { const int r = std::min(left, 138); cl_sym[ncl] = 18; cl_ext[ncl++] = (uint8_t)(r - 11); cf[18]++; left -= r; }
                if (left >= 3) { cl_sym[ncl] = 17; cl_ext[ncl++] = (uint8_t)(left - 3); cf[17]++; left = 0; }
                while (left--) { cl_sym[ncl] = 0; cl_ext[ncl++] = 0; cf[0]++; }
            } else {
                cl_sym[ncl] = (uint8_t)v; cl_ext[ncl++] = 0; cf[v]++; --left;
                while (left >= 3) { const int r = std::min(left, 6); cl_sym[ncl] = 16; cl_ext[ncl++] = (uint8_t)(r - 3); cf[16]++; left -= r; }
                while (left-- > 0) { cl_sym[ncl] = (uint8_t)v; cl_ext[ncl++] = 0; cf[v]++; }
            }
            i += run;
        }
        uint8_t cll[19];
        uint16_t clc[19];
        code_lengths(cf, 19, 7, cll);
        make_codes(cll, 19, clc);
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && cll[order[hclen - 1]] == 0) --hclen;
        put(final ? 1 : 0, 1);
        put(2, 2);
        put((uint32_t)(nlit - 257), 5);
        put((uint32_t)(ndist - 1), 5);
        put((uint32_t)(hclen - 4), 4);
        drain();
        for (int i = 0; i < hclen; ++i) { put(cll[order[i]], 3); if ((i & 7) == 7) drain(); }
        drain();
        for (int i = 0; i < ncl; ++i) {
            const int s = cl_sym[i];
            put(clc[s], cll[s]);
            if (s == 16) put(cl_ext[i], 2);
            else if (s == 17) put(cl_ext[i], 3);
            else if (s == 18) put(cl_ext[i], 7);
            drain();
        }
        // the symbols: a match is at most 15 + 5 + 15 + 13 = 48 bits; literals go out K per drain, K * (longest literal
        // code) + 7 <= 64: five at a time when no literal code is longer than 11 bits (FASTQ: the rule), else three
        uint32_t le[256];                                  // literal -> code | length << 16
        int maxlit = 0;
        for (int c = 0; c < 256; ++c) { le[c] = (uint32_t)lc[c] | ((uint32_t)ll[c] << 16); maxlit = std::max<int>(maxlit, ll[c]); }
        const bool five = maxlit <= 11;
        for (int i = 0; i < nsym; ++i) {
            const Sym s = syms_[i];
            if (s.dist == 0) {
                const uint8_t *p = src, *const end = src + s.litlen;
                drain();
                if (five) {
                    while (end - p >= 5) {
                        const uint32_t a = le[p[0]], b = le[p[1]], c = le[p[2]], d = le[p[3]], e = le[p[4]];
                        put(a & 0xFFFF, (int)(a >> 16));
                        put(b & 0xFFFF, (int)(b >> 16));
                        put(c & 0xFFFF, (int)(c >> 16));
                        put(d & 0xFFFF, (int)(d >> 16));
                        put(e & 0xFFFF, (int)(e >> 16));
                        drain();
                        p += 5;
                    }
                } else {
                    while (end - p >= 3) {
                        const uint32_t a = le[p[0]], b = le[p[1]], c = le[p[2]];
                        put(a & 0xFFFF, (int)(a >> 16));
                        put(b & 0xFFFF, (int)(b >> 16));
                        put(c & 0xFFFF, (int)(c >> 16));
                        drain();
                        p += 3;
                    }
                }
                while (p < end) { const uint32_t a = le[*p++]; put(a & 0xFFFF, (int)(a >> 16)); if (bitcnt_ > 40) drain(); }
                drain();
                src = end;
            } else {
                const int lcode = t.len_code[s.litlen], dcode = dist_sym(s.dist);
                put(lc[257 + lcode], ll[257 + lcode]);
                put((uint32_t)(s.litlen - t.len_base[lcode]), t.len_extra[lcode]);
                drain();
                put(dc[dcode], dl[dcode]);
                put((uint32_t)(s.dist - t.dist_base[dcode]), t.dist_extra[dcode]);
                drain();
                src += s.litlen;
            }
        }
        drain();
        put(lc[256], ll[256]);
        drain();
    }

    // ---- FASTQ records: what repeats inside 32 KiB of FASTQ is the text of the name lines (and of '+' lines that repeat
    // the name) -- the same characters at the same place one record earlier -- while bases and qualities are literals for
    // any LZ77 that does not search far.  So: no hash table; the name and '+' lines are compared with the record before at
    // the distance between the two line starts (runs of >= 4 equal bytes become matches), everything else goes out as
    // literal runs.  Same block / Huffman machinery as deflate_all, about the same ratio on FASTQ, no search cost.  Text
    // that is not FASTQ after all only compresses worse; the stream stays valid.
    void deflate_fastq(const uint8_t *in, size_t n) {
        const Tables &t = tab();
        int nsym = 0;
        size_t lit_from = 0, block_from = 0;
        auto literals = [&](size_t upto) {
            while (lit_from < upto) {
                const size_t k = std::min<size_t>(upto - lit_from, 65535);
                size_t j = lit_from;
                const size_t e4 = lit_from + (k & ~(size_t)3);
                for (; j < e4; j += 4) { h4_[0][in[j]]++; h4_[1][in[j + 1]]++; h4_[2][in[j + 2]]++; h4_[3][in[j + 3]]++; }
                for (; j < lit_from + k; ++j) h4_[0][in[j]]++;
                push_sym(nsym, Sym{(uint16_t)k, 0});
                lit_from += k;
            }
        };
        auto maybe_block = [&](size_t pos) {
            if (nsym >= BLOCK_SYMS - 64 || pos - block_from >= BLOCK_BYTES) {
                literals(pos);
                write_block(in + block_from, nsym, false);
                nsym = 0;
                block_from = pos;
            }
        };
        // [at, at + len) against the same bytes `dist` earlier: matches for the runs of >= 4 equal bytes
        auto same_place = [&](size_t at, size_t len, size_t dist) {
            size_t k = 0;
            while (k < len) {
                const uint8_t *a = in + at + k, *b = a - dist;
                const size_t maxl = len - k;
                size_t l = 0;
                while (l + 8 <= maxl) {
                    const uint64_t x = load64(a + l) ^ load64(b + l);
                    if (x) { l += (size_t)(__builtin_ctzll(x) >> 3); goto counted; }
                    l += 8;
                }
                while (l < maxl && a[l] == b[l]) ++l;
            counted:
                if (l >= 4 && nsym < BLOCK_SYMS - 16) {    // (symbol buffer nearly full: the rest of the line stays literal)
                    literals(at + k);
                    size_t left = l;
                    while (left) {                           // (a rest of 1..3 bytes would not be a legal match: take it from the piece before)
                        size_t piece = std::min<size_t>(left, MAX_MATCH);
                        if (left - piece > 0 && left - piece < 4) piece = left - 4;
                        push_sym(nsym, Sym{(uint16_t)piece, (uint16_t)dist});
                        lf_[257 + t.len_code[piece]]++;
                        df_[dist_sym((uint32_t)dist)]++;
                        left -= piece;
                    }
                    k += l;
                    lit_from = at + k;
                } else {
                    k += l + 1;                              // a short run and the byte that differs stay literals
                }
            }
        };
        size_t i = 0, prev_name = 0, prev_plus = 0;
        bool have_prev = false;
        while (i < n) {
            const uint8_t *e1 = (const uint8_t *)memchr(in + i, '\n', n - i);
            if (!e1) break;
            const uint8_t *e2 = (const uint8_t *)memchr(e1 + 1, '\n', n - (size_t)(e1 + 1 - in));
            if (!e2) break;
            const uint8_t *e3 = (const uint8_t *)memchr(e2 + 1, '\n', n - (size_t)(e2 + 1 - in));
            if (!e3) break;
            const uint8_t *e4 = (const uint8_t *)memchr(e3 + 1, '\n', n - (size_t)(e3 + 1 - in));
            if (!e4) break;
            const size_t name = i, plus = (size_t)(e2 + 1 - in), next = (size_t)(e4 + 1 - in);
            if (have_prev) {
                if (name - prev_name <= WINDOW) same_place(name, (size_t)(e1 + 1 - in) - name, name - prev_name);
                const size_t pl = (size_t)(e3 + 1 - in) - plus;
                if (pl > 4 && plus - prev_plus <= WINDOW) same_place(plus, pl, plus - prev_plus);
            }
            prev_name = name;
            prev_plus = plus;
            have_prev = true;
            i = next;
            maybe_block(i);
        }
        literals(n);
        write_block(in + block_from, nsym, true);
    }

    // ---- greedy LZ77 over the whole input, one block per BLOCK_SYMS symbols
    void deflate_all(const uint8_t *in, size_t n) {
        if (n == 0) {                                      // an empty stored block
            put(1, 1); put(0, 2);
            flush_bits();
            *op_++ = 0; *op_++ = 0; *op_++ = 0xFF; *op_++ = 0xFF;
            return;
        }
        if (epoch_ > 0xC0000000u || n > 0x3FFFFFF0u) { std::fill(head_.begin(), head_.end(), 0u); epoch_ = 0; }
        const uint32_t ep = epoch_;                        // head_ holds ep + position + 1; anything <= ep is from an earlier member
        epoch_ += (uint32_t)n + 1;
        const Tables &t = tab();
        int nsym = 0;
        size_t i = 0, lit_from = 0, block_from = 0;        // in[lit_from, i): literals not yet turned into a run symbol
        uint32_t miss = 0;
        const size_t safe = n >= 16 ? n - 16 : 0;          // positions below this can be hashed and matched with wide loads
        auto literals = [&](size_t upto) {                 // run symbols of at most 65535 bytes + their counts
            while (lit_from < upto) {
                const size_t k = std::min<size_t>(upto - lit_from, 65535);
                {   // four interleaved histograms: a run of equal bytes does not serialise on one counter
                    size_t j = lit_from;
                    const size_t e4 = lit_from + (k & ~(size_t)3);
                    for (; j < e4; j += 4) { h4_[0][in[j]]++; h4_[1][in[j + 1]]++; h4_[2][in[j + 2]]++; h4_[3][in[j + 3]]++; }
                    for (; j < lit_from + k; ++j) h4_[0][in[j]]++;
                }
                push_sym(nsym, Sym{(uint16_t)k, 0});
                lit_from += k;
            }
        };
        auto maybe_block = [&](size_t pos) {               // pos: everything in front of it is in symbols
            if (nsym >= BLOCK_SYMS || pos - block_from >= BLOCK_BYTES) {
                write_block(in + block_from, nsym, false);
                nsym = 0;
                block_from = pos;
            }
        };
        while (i < safe) {
            const uint32_t v = load32(in + i), h = hash4(v);
            const uint32_t stored = head_[h];
            head_[h] = ep + (uint32_t)(i + 1);
            const uint32_t cand = stored > ep ? stored - ep : 0u;          // position + 1 within this member, 0: none
            if (cand && i + 1 - cand <= WINDOW && load32(in + cand - 1) == v) {
                const uint8_t *a = in + i, *b = in + cand - 1;
                const size_t maxl = std::min<size_t>(MAX_MATCH, n - i);
                size_t l = 4;
                while (l + 8 <= maxl) {
                    const uint64_t x = load64(a + l) ^ load64(b + l);
                    if (x) { l += (size_t)(__builtin_ctzll(x) >> 3); goto extended; }
                    l += 8;
                }
                while (l < maxl && a[l] == b[l]) ++l;
            extended:
                if (l > maxl) l = maxl;
                literals(i);
                const uint32_t dist = (uint32_t)(i + 1 - cand);
                push_sym(nsym, Sym{(uint16_t)l, (uint16_t)dist});
                lf_[257 + t.len_code[l]]++;
                df_[dist_sym(dist)]++;
                // keep the table warm inside the match (every other position: cheap, and FASTQ repeats are long)
                const size_t end = i + l;
                for (size_t k = i + 2; k < end && k < safe; k += 2) head_[hash4(load32(in + k))] = ep + (uint32_t)(k + 1);
                i = end;
                lit_from = i;
                miss = 0;
                maybe_block(i);
                continue;
            }
            // no match here: after 16 misses in a row the scan starts to stride (random-looking stretches -- bases,
            // unbinned qualities -- cost a fraction of a probe per byte; the next hit resets the stride)
            i += 1 + (miss++ >> SNK_DEFLATE_MISS_SHIFT);
            if (i - lit_from >= LIT_FLUSH) {
                const size_t upto = std::min(i, safe);
                literals(upto);
                maybe_block(upto);
            }
        }
        literals(n);                                       // the tail
        write_block(in + block_from, nsym, true);
    }
};

}  // namespace snk
#endif
