// snk_inflate.h -- gzip (RFC 1952) / DEFLATE (RFC 1951) decoder for the FASTQ reader threads.
//
// The reference reads .gz input through zlib's gzgets (src/peprocess.cpp:2089): one inflate stream
// per file, ~0.3 GB/s of text per thread, which is the whole wall clock of this CLI for .gz input
// (DESIGN 4.1).  This decoder is the usual fast-path design -- 64-bit bit buffer refilled with one
// unaligned load, 11-bit / 8-bit primary Huffman tables with subtables, literal and match emitted
// straight into the caller's buffer, 8-byte match copies -- and decodes the whole compressed file from
// memory (the caller maps it).  Multi-member files are handled; the CRC-32 and ISIZE of every member
// are verified (zlib's crc32()).  It produces exactly the bytes zlib produces (tests: byte comparison
// with gzread on levels 1-9, stored blocks, multi-member and empty members).
//
// Contract of run(): `out` must be preceded by the previously produced bytes of the stream (at least
// the last 32 KiB of them, or all of them when fewer): matches reach back into them.
#ifndef SNK_INFLATE_H
#define SNK_INFLATE_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <zlib.h>
#include "snk_crc32.h"
#include <vector>

namespace snk {

class GzipInflate {
public:
    enum { HIST = 32768 };
    // a member that ended inside the output of the last run(): where, and what its trailer promises
    struct MemberEnd { size_t out_off; uint32_t crc, isize; };
    // verify = false: the caller checks the CRC-32 itself (e.g. on another thread) from member_ends()
    void set_verify_crc(bool v) { verify_crc_ = v; }
    const std::vector<MemberEnd> &member_ends() const { return ends_; }
    void init(const uint8_t *in, size_t n) {
        base_ = in;
        in_ = in; in_end_ = in + n;
        bitbuf_ = 0; bitcnt_ = 0;
        state_ = MEMBER_HEADER;
        err_ = nullptr;
        total_out_ = 0;
        member_out_ = 0;
        crc_ = 0;
        pend_len_ = 0;
        members_ = 0;
        partial_member_ = false;
        stop_bit_ = ~0ull;
        stopped_ = false;
    }
    // ---- mid-stream use (snk_pgunzip.h): start at the block header at bit offset `bit` of the input given to init();
    // the member it belongs to started earlier, so its length is unknown (no distance / ISIZE checks for it) and its
    // CRC is the caller's business (member_ends()).  The usual contract on `out` holds: the 32 KiB before it are the
    // stream's previous bytes.
    void start_at_block(uint64_t bit) {
        in_ = base_ + (bit >> 3);
        bitbuf_ = 0; bitcnt_ = 0;
        if (bit & 7) { refill(); take((int)(bit & 7)); }
        state_ = BLOCK_HEADER;
        member_out_ = 1ull << 40;
        partial_member_ = true;
        members_ = 1;
        verify_crc_ = false;
    }
    // run() returns (stopped() true) in front of the first block header at or past this bit offset
    void set_stop_bit(uint64_t b) { stop_bit_ = b; stopped_ = false; }
    bool stopped() const { return stopped_; }
    // (with stopped()) the block header run() stopped in front of is the first of its member: nothing of the member is out yet
    bool at_member_start() const { return state_ == BLOCK_HEADER && member_out_ == 0; }
    uint64_t bitpos() const { return (uint64_t)(in_ - base_) * 8 - (uint64_t)bitcnt_; }
    void set_strict(bool v) { strict_ = v; }
    const char *error() const { return err_; }
    bool done() const { return state_ == DONE; }
    uint64_t total_out() const { return total_out_; }

    // decodes up to cap bytes to out; returns the number produced (0 with done() or error() set at the end)
    size_t run(uint8_t *out, size_t cap) {
        uint8_t *const out0 = out, *const out_end = out + cap;
        acc_from_ = out0;
        run_out0_ = out0;
        ends_.clear();
        while (!err_ && state_ != DONE) {
            if (state_ == MEMBER_TRAILER) {                 // (no output needed: runs even when the buffer is full)
                account(out);                               // the CRC needs everything produced so far
                member_trailer();
                continue;
            }
            if (state_ == MEMBER_HEADER) { member_header(); continue; }
            if (state_ == BLOCK_HEADER && bitpos() >= stop_bit_) { stopped_ = true; break; }
            if (out == out_end) break;
            if (state_ == BLOCK_HEADER) { block_header(); continue; }
            if (state_ == STORED) {
                size_t n = stored_left_;
                if (n > (size_t)(out_end - out)) n = (size_t)(out_end - out);
                if (n > (size_t)(in_end_ - in_)) { err_ = "truncated stored block"; break; }
                memcpy(out, in_, n);
                in_ += n; out += n; stored_left_ -= (uint32_t)n;
                if (!stored_left_) state_ = final_ ? MEMBER_TRAILER : BLOCK_HEADER;
                continue;
            }
            out = codes(out, out_end);                      // CODES
        }
        account(out);
        return (size_t)(out - out0);
    }

protected:
    enum State { MEMBER_HEADER, BLOCK_HEADER, STORED, CODES, MEMBER_TRAILER, DONE };
    const uint8_t *base_ = nullptr;
    bool partial_member_ = false, stopped_ = false, strict_ = false;
    uint64_t stop_bit_ = ~0ull;
    enum { LIT_BITS = 11, DIST_BITS = 8, T_LIT = 1, T_LEN = 2, T_EOB = 3, T_SUB = 4 };
    const uint8_t *in_ = nullptr, *in_end_ = nullptr;
    uint64_t bitbuf_ = 0;
    int bitcnt_ = 0;
    State state_ = DONE;
    const char *err_ = nullptr;
    bool final_ = false;
    uint32_t stored_left_ = 0;
    uint32_t pend_len_ = 0, pend_dist_ = 0;            // match cut by the end of the output buffer
    uint64_t total_out_ = 0, member_out_ = 0;
    uint32_t crc_ = 0;
    const uint8_t *acc_from_ = nullptr, *run_out0_ = nullptr;
    bool verify_crc_ = true;
    std::vector<MemberEnd> ends_;
    uint64_t members_ = 0;
    uint32_t lit_[(1 << LIT_BITS) + 1024], dist_[(1 << DIST_BITS) + 512];

    // CRC / counters of the bytes produced since the last account() of this run()
    void account(uint8_t *out) {
        const size_t n = (size_t)(out - acc_from_);
        if (n) {
            if (verify_crc_) crc_ = snk::crc32_fast(crc_, acc_from_, n);
            member_out_ += n;
            total_out_ += n;
        }
        acc_from_ = out;
    }
    inline uint64_t in_member(const uint8_t *out) const { return member_out_ + (uint64_t)(out - acc_from_); }

    // ---- bit input
    inline void refill() {
        if (in_end_ - in_ >= 8) {
            uint64_t w;
            memcpy(&w, in_, 8);
            bitbuf_ |= w << bitcnt_;
            in_ += (63 - bitcnt_) >> 3;
            bitcnt_ |= 56;
        } else {
            while (bitcnt_ <= 56 && in_ < in_end_) { bitbuf_ |= (uint64_t)*in_++ << bitcnt_; bitcnt_ += 8; }
        }
    }
    inline bool need(int n) {                               // at least n bits available (after a refill)
        if (bitcnt_ < n) refill();
        return bitcnt_ >= n;
    }
    inline uint32_t take(int n) {
        const uint32_t v = (uint32_t)(bitbuf_ & ((1ull << n) - 1));
        bitbuf_ >>= n; bitcnt_ -= n;
        return v;
    }
    void byte_align() {                                     // drop to a byte boundary and give whole bytes back
        const int drop = bitcnt_ & 7;
        bitbuf_ >>= drop; bitcnt_ -= drop;
        in_ -= bitcnt_ >> 3;
        bitbuf_ = 0; bitcnt_ = 0;
    }

    bool member_header() {
        byte_align();
        if (in_ == in_end_) { state_ = DONE; return false; }
        // the magic first (zlib's gz_look): behind a member anything that is not 1F 8B -- a single byte included -- is trailing
        // garbage and gzread stops quietly, however few bytes there are; only a member that BEGINS and cannot be complete is truncated
        // (the length test used to come first: fewer than 18 bytes of padding were an error, 18 or more were not -- ADVICE r5)
        if (in_end_ - in_ < 2 || in_[0] != 0x1f || in_[1] != 0x8b) {
            if (members_) { state_ = DONE; return false; }
            err_ = in_end_ - in_ < 2 ? "truncated gzip member" : "not in gzip format"; return false;
        }
        if (in_end_ - in_ < 18) { err_ = "truncated gzip member"; return false; }
        if (in_[2] != 8) { err_ = "unknown gzip compression method"; return false; }
        const int flg = in_[3];
        const uint8_t *p = in_ + 10;
        if (flg & 4) {                                      // FEXTRA
            if (in_end_ - p < 2) { err_ = "truncated gzip header"; return false; }
            const size_t xl = p[0] | (p[1] << 8);
            p += 2;
            if ((size_t)(in_end_ - p) < xl) { err_ = "truncated gzip header"; return false; }
            p += xl;
        }
        for (int bit = 8; bit <= 16; bit <<= 1)             // FNAME, FCOMMENT
            if (flg & bit) {
                while (p < in_end_ && *p) ++p;
                if (p == in_end_) { err_ = "truncated gzip header"; return false; }
                ++p;
            }
        if (flg & 2) p += 2;                                // FHCRC
        if (p > in_end_) { err_ = "truncated gzip header"; return false; }
        in_ = p;
        crc_ = 0;
        member_out_ = 0;
        state_ = BLOCK_HEADER;
        return true;
    }
    bool member_trailer() {
        byte_align();
        if (in_end_ - in_ < 8) { err_ = "truncated gzip trailer"; return false; }
        const uint32_t crc = (uint32_t)in_[0] | ((uint32_t)in_[1] << 8) | ((uint32_t)in_[2] << 16) | ((uint32_t)in_[3] << 24);
        const uint32_t isz = (uint32_t)in_[4] | ((uint32_t)in_[5] << 8) | ((uint32_t)in_[6] << 16) | ((uint32_t)in_[7] << 24);
        in_ += 8;
        ++members_;
        if (verify_crc_) { if (crc != crc_) { err_ = "gzip CRC mismatch"; return false; } }
        else ends_.push_back(MemberEnd{(size_t)(acc_from_ - run_out0_), crc, isz});
        if (!partial_member_ && isz != (uint32_t)member_out_) { err_ = "gzip length mismatch"; return false; }
        partial_member_ = false;
        state_ = MEMBER_HEADER;
        return true;
    }

    // ---- Huffman tables
    static uint32_t rev(uint32_t c, int n) {
        uint32_t r = 0;
        for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1); c >>= 1; }
        return r;
    }
    // entry = value << 16 | extra << 12 | type << 8 | code length; sym_entry(sym) gives the payload (without length)
    // strict_ (block-start search): codes must be complete as zlib demands -- literal/length and code-length codes
    // always, distance codes unless there is a single code or none
    template <class F>
    bool build(uint32_t *tab, int tab_cap, int pbits, const uint8_t *lens, int nsym, F sym_entry, bool dist_code = false) {
        int count[16] = {0};
        for (int s = 0; s < nsym; ++s) count[lens[s]]++;
        count[0] = 0;
        int used = 0, maxlen = 0;
        for (int l = 1; l <= 15; ++l) if (count[l]) { used += count[l]; maxlen = l; }
        for (int i = 0; i < (1 << pbits); ++i) tab[i] = 0;  // invalid
        if (!used) return true;                              // no codes at all (allowed for distances)
        uint32_t next[16];
        uint32_t code = 0;
        long left = 1;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return false;                     // over-subscribed
            code = (code + (l > 1 ? (uint32_t)count[l - 1] : 0)) << 1;
            next[l] = code;
        }
        if (strict_ && left > 0 && !(dist_code && used == 1)) return false;   // incomplete
        // (incomplete codes are legal only for a single distance code; tolerated: unused entries stay invalid)
        // subtable sizes: per primary prefix the longest code
        int sub_bits[1 << LIT_BITS];
        if (maxlen > pbits) {
            for (int i = 0; i < (1 << pbits); ++i) sub_bits[i] = 0;
            uint32_t nx[16];
            for (int l = 1; l <= 15; ++l) nx[l] = next[l];
            for (int s = 0; s < nsym; ++s) {
                const int l = lens[s];
                if (!l) continue;
                const uint32_t c = nx[l]++;
                if (l > pbits) {
                    const uint32_t pre = rev(c, l) & ((1u << pbits) - 1);
                    if (l - pbits > sub_bits[pre]) sub_bits[pre] = l - pbits;
                }
            }
            int pool = 1 << pbits;
            for (int i = 0; i < (1 << pbits); ++i)
                if (sub_bits[i]) {
                    if (pool + (1 << sub_bits[i]) > tab_cap) return false;
                    tab[i] = ((uint32_t)pool << 16) | ((uint32_t)sub_bits[i] << 12) | (T_SUB << 8);
                    for (int k = 0; k < (1 << sub_bits[i]); ++k) tab[pool + k] = 0;
                    pool += 1 << sub_bits[i];
                }
        }
        for (int s = 0; s < nsym; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t c = rev(next[l]++, l);
            const uint32_t e = sym_entry(s) | (uint32_t)l;
            if (l <= pbits) {
                for (uint32_t i = c; i < (1u << pbits); i += 1u << l) tab[i] = e;
            } else {
                const uint32_t pe = tab[c & ((1u << pbits) - 1)];
                const int sb = (int)((pe >> 12) & 15);
                uint32_t *sub = tab + (pe >> 16);
                for (uint32_t i = c >> pbits; i < (1u << sb); i += 1u << (l - pbits)) sub[i] = e;
            }
        }
        return true;
    }
    static uint32_t litlen_entry(int s) {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) return ((uint32_t)s << 16) | (T_LIT << 8);
        if (s == 256) return T_EOB << 8;
        if (s > 285) return 0;                               // 286, 287: never valid in data
        return ((uint32_t)base[s - 257] << 16) | ((uint32_t)extra[s - 257] << 12) | (T_LEN << 8);
    }
    static uint32_t dist_entry(int s) {
        static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (s > 29) return 0;
        return ((uint32_t)base[s] << 16) | ((uint32_t)extra[s] << 12) | (T_LEN << 8);
    }

    bool block_header() {
        if (!need(3)) { err_ = "truncated deflate stream"; return false; }
        final_ = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 0) {
            byte_align();
            if (in_end_ - in_ < 4) { err_ = "truncated stored block"; return false; }
            const uint32_t len = in_[0] | (in_[1] << 8), nlen = in_[2] | (in_[3] << 8);
            if ((len ^ 0xFFFF) != nlen) { err_ = "invalid stored block lengths"; return false; }
            in_ += 4;
            stored_left_ = len;
            state_ = len ? STORED : (final_ ? MEMBER_TRAILER : BLOCK_HEADER);
            return true;
        }
        uint8_t lens[320];
        int nlit, ndist;
        if (type == 1) {
            nlit = 288; ndist = 32;
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else if (type == 2) {
            if (!need(14)) { err_ = "truncated deflate stream"; return false; }
            nlit = (int)take(5) + 257;
            ndist = (int)take(5) + 1;
            const int ncl = (int)take(4) + 4;
            if (nlit > 286 || ndist > 30) { err_ = "too many length or distance symbols"; return false; }
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncl; ++i) {
                if (!need(3)) { err_ = "truncated deflate stream"; return false; }
                cl[order[i]] = (uint8_t)take(3);
            }
            uint32_t ctab[1 << 7];
            if (!build(ctab, 1 << 7, 7, cl, 19, [](int s) { return ((uint32_t)s << 16) | (T_LIT << 8); })) { err_ = "invalid code lengths set"; return false; }
            int i = 0;
            while (i < nlit + ndist) {
                refill();
                const uint32_t e = ctab[bitbuf_ & 127];
                if (!(e & 0xFF) || (int)(e & 0xFF) > bitcnt_) { err_ = "invalid code lengths set"; return false; }
                take((int)(e & 0xFF));
                const int sym = (int)(e >> 16);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                const int xb = sym == 16 ? 2 : (sym == 17 ? 3 : 7);
                if (bitcnt_ < xb) { err_ = "truncated deflate stream"; return false; }
                if (sym == 16) {
                    if (i == 0) { err_ = "invalid bit length repeat"; return false; }
                    val = lens[i - 1];
                    rep = 3 + (int)take(2);
                } else if (sym == 17) rep = 3 + (int)take(3);
                else rep = 11 + (int)take(7);
                if (i + rep > nlit + ndist) { err_ = "invalid bit length repeat"; return false; }
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) { err_ = "invalid code -- missing end-of-block"; return false; }
            // the two alphabets sit back to back in lens[]: move the distance lengths to a fixed place
            uint8_t dl[32];
            for (int k = 0; k < ndist; ++k) dl[k] = lens[nlit + k];
            for (int k = 0; k < ndist; ++k) lens[288 + k] = dl[k];
            for (int k = nlit; k < 288; ++k) lens[k] = 0;
            for (int k = ndist; k < 32; ++k) lens[288 + k] = 0;
            nlit = 288; ndist = 32;
        } else {
            err_ = "invalid block type";
            return false;
        }
        if (!build(lit_, (int)(sizeof(lit_) / 4), LIT_BITS, lens, nlit, litlen_entry)) { err_ = "invalid literal/lengths set"; return false; }
        if (!build(dist_, (int)(sizeof(dist_) / 4), DIST_BITS, lens + 288, ndist, dist_entry, true)) { err_ = "invalid distances set"; return false; }
        state_ = CODES;
        return true;
    }

    static inline void copy_match(uint8_t *out, uint32_t dist, uint32_t len) {   // may write up to 7 bytes past len
        const uint8_t *src = out - dist;
        if (dist >= 8) {
            uint8_t *const end = out + len;
            do { uint64_t w; memcpy(&w, src, 8); memcpy(out, &w, 8); src += 8; out += 8; } while (out < end);
        } else if (dist == 1) {
            memset(out, *src, len);
        } else {
            for (uint32_t i = 0; i < len; ++i) out[i] = src[i];
        }
    }

    // Huffman-coded data of one block; returns the new output position
    uint8_t *codes(uint8_t *out, uint8_t *out_end) {
        if (pend_len_) {                                     // finish the match the previous call had to cut
            uint32_t n = pend_len_;
            if (n > (uint32_t)(out_end - out)) n = (uint32_t)(out_end - out);
            for (uint32_t i = 0; i < n; ++i) out[i] = out[(ptrdiff_t)i - (ptrdiff_t)pend_dist_];
            out += n;
            pend_len_ -= n;
            if (pend_len_) return out;
        }
        const uint32_t lmask = (1u << LIT_BITS) - 1, dmask = (1u << DIST_BITS) - 1;
        for (;;) {
            // fast path: room for the longest match plus the copy overrun, input for a whole symbol pair.
            // The bit buffer lives in locals: the byte stores to `out` may alias the members and would
            // force them through memory on every literal.
            if (out_end - out >= 258 + 8 && in_end_ - in_ >= 16) {
                uint64_t bb = bitbuf_;
                int bc = bitcnt_;
                const uint8_t *in = in_;
                const uint8_t *const in_lim = in_end_ - 16;
                uint8_t *const out_lim = out_end - (258 + 8);
                const uint64_t member_base = member_out_ - (uint64_t)(acc_from_ - out);   // bytes of the member before `out`
                uint8_t *const out_base = out;
                const char *bad = nullptr;
                bool eob = false;
                while (out <= out_lim && in <= in_lim) {
                    uint64_t w;
                    memcpy(&w, in, 8);
                    bb |= w << bc;
                    in += (63 - bc) >> 3;
                    bc |= 56;
                    uint32_t e = lit_[bb & lmask];
                    if (__builtin_expect(((e >> 8) & 15) == T_SUB, 0)) e = lit_[(e >> 16) + ((bb >> LIT_BITS) & ((1u << ((e >> 12) & 15)) - 1))];
                    uint32_t t = (e >> 8) & 15;
                    bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                    if (t == T_LIT) {                       // up to three literals per refill (3 x 15 bits <= 56)
                        *out++ = (uint8_t)(e >> 16);
                        e = lit_[bb & lmask];
                        if (((e >> 8) & 15) != T_LIT) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *out++ = (uint8_t)(e >> 16);
                        e = lit_[bb & lmask];
                        if (((e >> 8) & 15) != T_LIT) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *out++ = (uint8_t)(e >> 16);
                        continue;
                    }
                    if (__builtin_expect(t == T_LEN, 1)) {
                        const int xb = (int)((e >> 12) & 15);
                        const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xb) - 1));
                        bb >>= xb; bc -= xb;
                        uint32_t d = dist_[bb & dmask];
                        if (__builtin_expect(((d >> 8) & 15) == T_SUB, 0)) d = dist_[(d >> 16) + ((bb >> DIST_BITS) & ((1u << ((d >> 12) & 15)) - 1))];
                        if (__builtin_expect(((d >> 8) & 15) != T_LEN, 0)) { bad = "invalid distance code"; break; }
                        bb >>= (d & 0xFF); bc -= (int)(d & 0xFF);
                        const int dxb = (int)((d >> 12) & 15);
                        const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << dxb) - 1));
                        bb >>= dxb; bc -= dxb;
                        if (__builtin_expect((uint64_t)dist > member_base + (uint64_t)(out - out_base), 0)) { bad = "invalid distance too far back"; break; }
                        copy_match(out, dist, len);
                        out += len;
                        continue;
                    }
                    if (t == T_EOB) { eob = true; break; }
                    bad = "invalid literal/length code";
                    break;
                }
                bitbuf_ = bb; bitcnt_ = bc; in_ = in;
                if (bad) { err_ = bad; return out; }
                if (eob) { state_ = final_ ? MEMBER_TRAILER : BLOCK_HEADER; return out; }
                if (out_end - out >= 258 + 8 && in_end_ - in_ >= 16) continue;      // (cannot happen; keeps the invariant obvious)
            }
            // careful path (ends of the buffers): one symbol at a time, everything checked
            if (out == out_end) return out;
            refill();
            uint32_t e = lit_[bitbuf_ & lmask];
            if (((e >> 8) & 15) == T_SUB) e = lit_[(e >> 16) + ((bitbuf_ >> LIT_BITS) & ((1u << ((e >> 12) & 15)) - 1))];
            const uint32_t t = (e >> 8) & 15;
            if (t == 0 || (int)(e & 0xFF) > bitcnt_) { err_ = bitcnt_ < 15 && in_ == in_end_ ? "truncated deflate stream" : "invalid literal/length code"; return out; }
            bitbuf_ >>= (e & 0xFF); bitcnt_ -= (int)(e & 0xFF);
            if (t == T_LIT) { *out++ = (uint8_t)(e >> 16); continue; }
            if (t == T_EOB) { state_ = final_ ? MEMBER_TRAILER : BLOCK_HEADER; return out; }
            const int xb = (int)((e >> 12) & 15);
            if (bitcnt_ < xb) { err_ = "truncated deflate stream"; return out; }
            uint32_t len = (e >> 16) + (uint32_t)(bitbuf_ & ((1u << xb) - 1));
            bitbuf_ >>= xb; bitcnt_ -= xb;
            refill();
            uint32_t d = dist_[bitbuf_ & dmask];
            if (((d >> 8) & 15) == T_SUB) d = dist_[(d >> 16) + ((bitbuf_ >> DIST_BITS) & ((1u << ((d >> 12) & 15)) - 1))];
            if (((d >> 8) & 15) != T_LEN || (int)(d & 0xFF) > bitcnt_) { err_ = "invalid distance code"; return out; }
            bitbuf_ >>= (d & 0xFF); bitcnt_ -= (int)(d & 0xFF);
            const int dxb = (int)((d >> 12) & 15);
            if (bitcnt_ < dxb) { err_ = "truncated deflate stream"; return out; }
            const uint32_t dist = (d >> 16) + (uint32_t)(bitbuf_ & ((1u << dxb) - 1));
            bitbuf_ >>= dxb; bitcnt_ -= dxb;
            if ((uint64_t)dist > in_member(out)) { err_ = "invalid distance too far back"; return out; }
            uint32_t n = len;
            if (n > (uint32_t)(out_end - out)) n = (uint32_t)(out_end - out);
            for (uint32_t i = 0; i < n; ++i) out[i] = out[(ptrdiff_t)i - (ptrdiff_t)dist];
            out += n;
            if (n < len) { pend_len_ = len - n; pend_dist_ = dist; return out; }
        }
    }
};

}  // namespace snk
#endif
