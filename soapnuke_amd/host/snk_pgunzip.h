// snk_pgunzip.h -- parallel decompression of ONE gzip stream ("parallel inflate ... the real speed lever").
//
// The reference reads .gz input through zlib's gzgets (src/peprocess.cpp:2089-2113): one inflate stream per file.  A
// DEFLATE stream has no index, but it can still be decoded from the middle (the two-pass scheme of pugz / rapidgzip,
// restated here from its published description):
//
//   1. The compressed file is cut into chunks of a few MB.  For every chunk but the first a worker SEARCHES the first
//      deflate block that starts at or after the chunk's first byte: a candidate bit offset must hold a non-final
//      dynamic-Huffman header whose three codes are complete (what zlib itself demands), the block must decode to its
//      end-of-block symbol and be followed by another valid header.
//   2. The worker decodes from there WITHOUT knowing the 32 KiB of text in front of it: output symbols are 16 bit wide,
//      a value >= 256 is a MARKER "byte number (v - 256) of the unknown window".  Markers are copied like literals by
//      later matches (in FASTQ they never die out: every read name copies its prefix from the one before).  When a gzip
//      member ends, the rest of the chunk is decoded by the ordinary byte decoder (snk_inflate.h): nothing reaches back
//      across a member boundary.  A chunk ends exactly at the block where a later chunk starts (found starts that are
//      never hit are false positives and simply skipped).
//   3. A chain thread walks the finished chunks in stream order.  It knows the real window in front of a chunk (the last
//      32 KiB of everything before), resolves just the chunk's last 32 KiB -- the next chunk's window -- and hands the
//      chunk back to the workers, which replace all its markers and compute the CRC-32 of its pieces in parallel.
//   4. The consumer copies the resolved chunks out in order, chains the CRCs (crc32_combine) and checks every member
//      trailer.  Anything that does not fit -- a chunk that failed, a chain that does not meet -- falls back to plain
//      sequential decoding from that point on: the result is always what zlib would produce, or an error.
#ifndef SNK_PGUNZIP_H
#define SNK_PGUNZIP_H
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "snk_inflate.h"

namespace snk {

// DEFLATE decoder with an unknown window: 16-bit output symbols, markers for references into the window
class MarkerInflate : public GzipInflate {
public:
    enum Why { STOP, MEMBER_END, FULL, FAIL };
    // Decodes whole blocks from the current block header (start_at_block()) into M[0, len) until: the position is at
    // or past stop_bit (STOP), a final block ended (MEMBER_END: the trailer is next), len would exceed max_syms (FULL),
    // or the data are invalid (FAIL).  M grows as needed (its size is a capacity; len counts the symbols).
    Why run_markers(std::vector<uint16_t> &M, size_t &len, uint64_t stop_bit, size_t max_syms) {
        for (;;) {
            if (bitpos() >= stop_bit) return STOP;
            const Why w = one_block(M, len, max_syms);
            if (w != STOP) return w;
            if (state_ == MEMBER_TRAILER) return MEMBER_END;
        }
    }
    // block-start probe: exactly one (dynamic) block from the current header ...
    Why run_markers_one(std::vector<uint16_t> &M, size_t max_syms) {
        size_t len = 0;
        const Why w = one_block(M, len, max_syms);
        if (w != STOP) return w;
        return state_ == MEMBER_TRAILER ? MEMBER_END : STOP;
    }
    // ... and the header behind it
    bool header_ok_after() { return block_header(); }

private:
    static void room(std::vector<uint16_t> &M, size_t need) {
        if (M.size() < need) M.resize(std::max(need, M.size() + M.size() / 2 + 65536));
    }
    Why one_block(std::vector<uint16_t> &M, size_t &len, size_t max_syms) {
        if (!block_header()) return FAIL;
        if (state_ == STORED) {
            if ((size_t)(in_end_ - in_) < stored_left_) return FAIL;
            if (len + stored_left_ > max_syms) return FULL;
            room(M, len + stored_left_);
            for (uint32_t i = 0; i < stored_left_; ++i) M[len + i] = in_[i];
            len += stored_left_;
            in_ += stored_left_;
            stored_left_ = 0;
            state_ = final_ ? MEMBER_TRAILER : BLOCK_HEADER;
            return STOP;
        }
        return state_ == CODES ? codes16(M, len, max_syms) : STOP;
    }
    static inline void copy16(uint16_t *o, uint32_t dist, uint32_t len) {      // may write up to 3 symbols past len
        const uint16_t *src = o - dist;
        if (dist >= 4) {
            uint16_t *const end = o + len;
            do { uint64_t w; memcpy(&w, src, 8); memcpy(o, &w, 8); src += 4; o += 4; } while (o < end);
        } else {
            for (uint32_t i = 0; i < len; ++i) o[i] = src[i];
        }
    }
    // one block's symbols; STOP = end of block
    Why codes16(std::vector<uint16_t> &M, size_t &len, size_t max_syms) {
        const uint32_t lmask = (1u << LIT_BITS) - 1, dmask = (1u << DIST_BITS) - 1;
        for (;;) {
            if (len + 600 > max_syms) return FULL;
            room(M, len + (1 << 16));
            // fast path: the bit buffer in locals, room for 64 Ki symbols, input for a whole symbol pair
            if (in_end_ - in_ >= 16) {
                uint64_t bb = bitbuf_;
                int bc = bitcnt_;
                const uint8_t *in = in_;
                const uint8_t *const in_lim = in_end_ - 16;
                uint16_t *const base = M.data();
                uint16_t *o = base + len, *const o_lim = base + M.size() - 300;
                bool eob = false, bad = false;
                while (o <= o_lim && in <= in_lim) {
                    uint64_t w;
                    memcpy(&w, in, 8);
                    bb |= w << bc;
                    in += (63 - bc) >> 3;
                    bc |= 56;
                    uint32_t e = lit_[bb & lmask];
                    if (__builtin_expect(((e >> 8) & 15) == T_SUB, 0)) e = lit_[(e >> 16) + ((bb >> LIT_BITS) & ((1u << ((e >> 12) & 15)) - 1))];
                    uint32_t t = (e >> 8) & 15;
                    bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                    if (t == T_LIT) {
                        *o++ = (uint16_t)(e >> 16);
                        e = lit_[bb & lmask];
                        if (((e >> 8) & 15) != T_LIT) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *o++ = (uint16_t)(e >> 16);
                        e = lit_[bb & lmask];
                        if (((e >> 8) & 15) != T_LIT) continue;
                        bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
                        *o++ = (uint16_t)(e >> 16);
                        continue;
                    }
                    if (__builtin_expect(t == T_LEN, 1)) {
                        const int xb = (int)((e >> 12) & 15);
                        const uint32_t mlen = (e >> 16) + (uint32_t)(bb & ((1u << xb) - 1));
                        bb >>= xb; bc -= xb;
                        uint32_t d = dist_[bb & dmask];
                        if (__builtin_expect(((d >> 8) & 15) == T_SUB, 0)) d = dist_[(d >> 16) + ((bb >> DIST_BITS) & ((1u << ((d >> 12) & 15)) - 1))];
                        if (__builtin_expect(((d >> 8) & 15) != T_LEN, 0)) { bad = true; break; }
                        bb >>= (d & 0xFF); bc -= (int)(d & 0xFF);
                        const int dxb = (int)((d >> 12) & 15);
                        const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << dxb) - 1));
                        bb >>= dxb; bc -= dxb;
                        const size_t pos = (size_t)(o - base);
                        if (__builtin_expect(dist > pos, 0)) {          // starts inside the unknown window
                            const uint32_t before = (uint32_t)(dist - pos);
                            if (before > HIST) { bad = true; break; }
                            const uint32_t nb = before < mlen ? before : mlen;
                            for (uint32_t i = 0; i < nb; ++i) o[i] = (uint16_t)(256 + HIST - before + i);
                            for (uint32_t i = nb; i < mlen; ++i) o[i] = o[(ptrdiff_t)i - (ptrdiff_t)dist];
                        } else {
                            copy16(o, dist, mlen);
                        }
                        o += mlen;
                        continue;
                    }
                    if (t == T_EOB) { eob = true; break; }
                    bad = true;
                    break;
                }
                bitbuf_ = bb; bitcnt_ = bc; in_ = in;
                len = (size_t)(o - base);
                if (bad) return FAIL;
                if (eob) { state_ = final_ ? MEMBER_TRAILER : BLOCK_HEADER; return STOP; }
                if (in_end_ - in_ >= 16) continue;              // out of room: grow and go on
            }
            // careful path (end of the input): one symbol at a time, everything checked
            refill();
            uint32_t e = lit_[bitbuf_ & lmask];
            if (((e >> 8) & 15) == T_SUB) e = lit_[(e >> 16) + ((bitbuf_ >> LIT_BITS) & ((1u << ((e >> 12) & 15)) - 1))];
            const uint32_t t = (e >> 8) & 15;
            if (t == 0 || (int)(e & 0xFF) > bitcnt_) return FAIL;
            bitbuf_ >>= (e & 0xFF); bitcnt_ -= (int)(e & 0xFF);
            if (t == T_LIT) { M[len++] = (uint16_t)(e >> 16); continue; }
            if (t == T_EOB) { state_ = final_ ? MEMBER_TRAILER : BLOCK_HEADER; return STOP; }
            const int xb = (int)((e >> 12) & 15);
            if (bitcnt_ < xb) return FAIL;
            const uint32_t mlen = (e >> 16) + (uint32_t)(bitbuf_ & ((1u << xb) - 1));
            bitbuf_ >>= xb; bitcnt_ -= xb;
            refill();
            uint32_t d = dist_[bitbuf_ & dmask];
            if (((d >> 8) & 15) == T_SUB) d = dist_[(d >> 16) + ((bitbuf_ >> DIST_BITS) & ((1u << ((d >> 12) & 15)) - 1))];
            if (((d >> 8) & 15) != T_LEN || (int)(d & 0xFF) > bitcnt_) return FAIL;
            bitbuf_ >>= (d & 0xFF); bitcnt_ -= (int)(d & 0xFF);
            const int dxb = (int)((d >> 12) & 15);
            if (bitcnt_ < dxb) return FAIL;
            const uint32_t dist = (d >> 16) + (uint32_t)(bitbuf_ & ((1u << dxb) - 1));
            bitbuf_ >>= dxb; bitcnt_ -= dxb;
            if (dist > len + HIST) return FAIL;
            uint16_t *o = M.data() + len;
            for (uint32_t i = 0; i < mlen; ++i) {
                const ptrdiff_t k = (ptrdiff_t)len + (ptrdiff_t)i - (ptrdiff_t)dist;
                o[i] = k < 0 ? (uint16_t)(256 + HIST + k) : M[(size_t)k];
            }
            len += mlen;
        }
    }
};

class ParallelGunzip {
public:
    enum { HIST = GzipInflate::HIST };
    // start_bit / start_win: decode from the deflate block header at that bit offset on, the 32 KiB of text in front of it
    // given (a shard of a sharded run, host/snk_main.cpp: the parent's scout pass has decoded the stream once, verified every
    // member's CRC-32 and noted such places); the member the start lies in cannot be checked again and is not
    ParallelGunzip(const uint8_t *in, size_t n, int threads, size_t chunk_bytes = (size_t)2 << 20, uint64_t start_bit = ~0ull, const uint8_t *start_win = nullptr)
        : in_(in), n_(n), cb_(chunk_bytes < 65536 ? 65536 : chunk_bytes) {
        nchunks_ = (n_ + cb_ - 1) / cb_;
        if (nchunks_ == 0) nchunks_ = 1;
        chunks_.reset(new Chunk[nchunks_]);
        win_.assign(HIST, 0);
        if (start_bit != ~0ull && start_win && (start_bit >> 3) < n_) {
            mid_ = partial_ = true;
            start_bit_ = start_bit;
            first_ = (size_t)(start_bit >> 3) / cb_;
            win0_.assign(start_win, start_win + HIST);
            next_task_ = consumed_ = first_;
        }
        threads = threads < 1 ? 1 : threads;
        lookahead_ = (size_t)threads + threads / 2 + 2;
        if (nchunks_ - first_ > 2 && threads > 1) {
            for (int t = 0; t < threads; ++t) pool_.emplace_back([this] { worker(); });
            chain_thread_ = std::thread([this] { chain(); });
        } else if (mid_) {
            win_ = win0_;
            seq_from(start_bit_, false);
        } else {
            seq_from(0, true);                              // small input or one thread: plain sequential decoding
        }
    }
    // called from run() whenever the text of a chunk of the chain begins: the chunk's first block header (bit offset; 0 = the
    // gzip header of the stream) and the number of bytes run() has handed out before it -- a place a later decoder can start from
    std::function<void(uint64_t bit, uint64_t text_off)> on_chunk;
    ~ParallelGunzip() {
        { std::lock_guard<std::mutex> l(m_); quit_ = true; }
        cv_.notify_all();
        for (auto &t : pool_) t.join();
        if (chain_thread_.joinable()) chain_thread_.join();
    }
    const char *error() const { return err_; }
    bool done() const { return done_; }

    // the next bytes of the stream, in order; 0 at the end (done()) or after an error (error())
    size_t run(uint8_t *out, size_t cap) {
        size_t got = 0;
        while (got < cap && !done_ && !err_) {
            if (seq_) { got += seq_run(out + got, cap - got); continue; }
            if (!cur_ready_ && !next_chunk()) continue;
            Chunk &c = chunks_[cur_];
            const size_t n = std::min(cap - got, c.out_len - cur_off_);
            memcpy(out + got, c.buf.data() + HIST + cur_off_, n);
            got += n;
            handed_out_ += n;
            cur_off_ += n;
            if (cur_off_ == c.out_len) release_current();
        }
        return got;
    }

private:
    struct Piece { uint64_t len; uint32_t crc; bool ends_member; uint32_t want_crc, want_isize; };
    struct Chunk {
        std::atomic<int> searched{0};                      // 0 no, 1 in progress, 2 done
        uint64_t start_bit = ~0ull;                        // ~0: no block start found in this chunk
        std::atomic<int> done{0}, resolved{0};
        bool ok = false, whole_tail = false;               // whole_tail: ran to the end of the input
        uint64_t end_bit = 0;
        std::vector<uint16_t> M;                           // marker part (front of the chunk's output), mlen symbols
        size_t mlen = 0;
        std::vector<uint8_t> buf;                          // [HIST pad | resolved M | byte-decoded rest]
        size_t out_len = 0;
        std::vector<Piece> pieces;                         // the byte-decoded rest, cut at member ends
        std::vector<uint8_t> win;                          // the 32 KiB in front of this chunk (set by the chain thread)
        uint32_t crc_front = 0;                            // CRC-32 of the resolved marker part
    };

    const uint8_t *in_;
    size_t n_, cb_, nchunks_ = 0, lookahead_ = 4;
    bool mid_ = false, partial_ = false;                   // started inside the stream / inside a member whose front was not seen
    uint64_t start_bit_ = 0, handed_out_ = 0;
    size_t first_ = 0;                                     // the chunk the stream (or the part asked for) begins in
    std::vector<uint8_t> win0_;
    std::unique_ptr<Chunk[]> chunks_;
    std::vector<std::thread> pool_;
    std::thread chain_thread_;
    std::mutex m_;
    std::condition_variable cv_;
    bool quit_ = false;
    size_t next_task_ = 0, consumed_ = 0;                  // chunk indices: next to decode / all below are released
    size_t chain_need_ = 0;                                // the chain thread waits for this chunk (or an earlier one)
    std::vector<size_t> resolve_q_;                        // chunks waiting for their markers to be replaced (FIFO by position)
    size_t resolve_head_ = 0;
    // the chain: chunk indices in stream order, as far as the chain thread got
    std::vector<size_t> order_;
    bool chain_end_ = false, chain_fallback_ = false;      // the stream's end was reached / the chain broke (fb_*)
    uint64_t fb_bit_ = 0;
    bool fb_header_ = false;
    std::vector<uint8_t> fb_win_;
    // consumer state
    const char *err_ = nullptr;
    bool done_ = false, cur_ready_ = false;
    size_t cur_ = 0, cur_off_ = 0, order_pos_ = 0;
    std::vector<uint8_t> win_;                             // fallback: the last 32 KiB delivered
    uint32_t mcrc_ = 0;                                    // CRC-32 / length of the current member so far
    uint64_t mlen_ = 0;
    // sequential fallback
    bool seq_ = false;
    GzipInflate sq_;
    std::vector<uint8_t> sbuf_;                            // [HIST window | block]
    size_t s_have_ = 0, s_off_ = 0;

    // ---------------------------------------------------------------- block start search
    static inline uint64_t peek(const uint8_t *p, const uint8_t *end, uint64_t bit) {
        const uint8_t *q = p + (bit >> 3);
        uint64_t w = 0;
        if (end - q >= 8) memcpy(&w, q, 8);
        else for (int i = 0; q + i < end; ++i) w |= (uint64_t)q[i] << (8 * i);
        return w >> (bit & 7);
    }
    // first plausible block start in [from, to) (bit offsets), ~0 if none
    uint64_t find_block(uint64_t from, uint64_t to) const {
        const uint8_t *end = in_ + n_;
        MarkerInflate z;
        std::vector<uint16_t> scratch;
        for (uint64_t b = from; b < to; ++b) {
            const uint64_t h = peek(in_, end, b);
            if ((h & 7) != 4) continue;                     // BFINAL 0, BTYPE 10 (dynamic)
            if (((h >> 3) & 31) > 29 || ((h >> 8) & 31) > 29) continue;
            const int ncl = (int)((h >> 13) & 15) + 4;
            const uint64_t cl = peek(in_, end, b + 17);
            int kraft = 0, nz = 0;
            for (int i = 0; i < ncl; ++i) {
                const int l = (int)((cl >> (3 * i)) & 7);
                if (l) { kraft += 128 >> l; ++nz; }
            }
            if (kraft != 128 || nz < 2) continue;           // the code-length code must be complete
            // full check: this block and the header after it, decoded for real
            z.init(in_, n_);
            z.set_strict(true);
            z.start_at_block(b);
            const MarkerInflate::Why w = z.run_markers_one(scratch, (size_t)1 << 22);
            if (w != MarkerInflate::STOP) continue;         // failed, or "ended a member" although BFINAL was 0
            if (!z.header_ok_after()) continue;
            return b;
        }
        return ~0ull;
    }
    uint64_t ensure_search(size_t i) {
        Chunk &c = chunks_[i];
        const int s = c.searched.load(std::memory_order_acquire);
        if (s == 2) return c.start_bit;
        int zero = 0;
        if (s == 0 && c.searched.compare_exchange_strong(zero, 1)) {
            c.start_bit = i == first_ ? (mid_ ? start_bit_ : 0) : find_block((uint64_t)i * cb_ * 8, std::min<uint64_t>((uint64_t)(i + 1) * cb_ * 8, (uint64_t)n_ * 8));
            c.searched.store(2, std::memory_order_release);
            return c.start_bit;
        }
        while (c.searched.load(std::memory_order_acquire) != 2) std::this_thread::yield();   // another thread is on it
        return c.start_bit;
    }

    // ---------------------------------------------------------------- workers
    void worker() {
        for (;;) {
            size_t i;
            bool resolve = false;
            {
                std::unique_lock<std::mutex> l(m_);
                // decode tasks: inside the consumer's window, or wanted by the chain thread (it may have to walk past any number
                // of chunks without a matching block start -- single-block members, long stored runs -- before the consumer
                // gets anything to release: ADVICE r2, deadlock); none once the chain has ended or fallen back
                cv_.wait(l, [&] {
                    return quit_ || resolve_head_ < resolve_q_.size() ||
                           (!chain_end_ && !chain_fallback_ && next_task_ < nchunks_ &&
                            (next_task_ < consumed_ + lookahead_ || next_task_ <= chain_need_));
                });
                if (quit_) return;
                if (resolve_head_ < resolve_q_.size()) { i = resolve_q_[resolve_head_++]; resolve = true; }
                else i = next_task_++;
            }
            if (resolve) {
                resolve_chunk(chunks_[i]);
                chunks_[i].resolved.store(1, std::memory_order_release);
            } else {
                decode_chunk(i);
                chunks_[i].done.store(1, std::memory_order_release);
            }
            { std::lock_guard<std::mutex> l(m_); }
            cv_.notify_all();
        }
    }
    // the first found block start behind chunk j-1 at or past `bit`; the search gives up (gave_up = true) after SEARCH_AHEAD
    // chunks without one: a stream without non-final dynamic blocks (many small single-block members, stored data) has no
    // starts at all, and one thread walking find_block over the whole file is slower than decoding it
    enum { SEARCH_AHEAD = 6 };
    uint64_t next_stop(size_t &j, uint64_t bit, bool &gave_up) {
        gave_up = false;
        const size_t lim = std::min(nchunks_, j + (size_t)SEARCH_AHEAD);
        for (; j < lim; ++j) {
            {
                std::lock_guard<std::mutex> l(m_);
                if (quit_ || chain_fallback_) { gave_up = true; return ~0ull; }
            }
            const uint64_t s = ensure_search(j);
            if (s != ~0ull && s >= bit) return s;
        }
        gave_up = j < nchunks_;
        return ~0ull;
    }
    void decode_chunk(size_t i) {
        Chunk &c = chunks_[i];
        const uint64_t s0 = ensure_search(i);
        if (s0 == ~0ull) return;
        const size_t cap_syms = std::max<size_t>(cb_ * 64, (size_t)64 << 20);   // a runaway (false start) gives up here
        MarkerInflate z;
        z.init(in_, n_);
        z.set_verify_crc(false);
        size_t j = i + 1;
        bool gave_up = false;
        uint64_t stop = next_stop(j, s0 + 1, gave_up);
        if (gave_up) return;                                // no block start near: the chain breaks here, the consumer decodes sequentially
        bool byte_mode = i == first_;                       // the first chunk starts at the gzip header, or with its window given: nothing unknown
        if (i > first_) {
            z.start_at_block(s0);
            take(c.M);
            if (c.M.size() < cb_ * 3 + 65536) c.M.resize(cb_ * 3 + 65536);
        } else if (mid_) z.start_at_block(s0);
        // ---- marker phase
        while (!byte_mode) {
            const MarkerInflate::Why w = z.run_markers(c.M, c.mlen, stop, cap_syms);
            if (w == MarkerInflate::FAIL || w == MarkerInflate::FULL) return;
            if (w == MarkerInflate::STOP) {
                if (z.bitpos() == stop) { finish(c, z, 0, false); return; }
                ++j;                                        // ran past it: that start was a false positive
                stop = next_stop(j, z.bitpos(), gave_up);
                if (gave_up) return;
                continue;
            }
            byte_mode = true;                               // MEMBER_END: nothing reaches back across it
        }
        // ---- byte phase: [HIST pad][mlen bytes reserved for the resolved markers][decoded bytes]
        const size_t mlen = c.mlen;
        size_t cap = mlen + std::max<size_t>(cb_ * 4, (size_t)1 << 20);
        take(c.buf);
        if (c.buf.size() < cap + HIST) c.buf.resize(cap + HIST); else cap = c.buf.size() - HIST;
        if (i == first_ && mid_) memcpy(c.buf.data(), win0_.data(), HIST);       // (the decoder's contract: the 32 KiB in front of `out` are the stream's)
        uint8_t *base = c.buf.data() + HIST;
        size_t pos = mlen, piece_from = mlen;
        uint32_t crc = 0;
        for (;;) {
            z.set_stop_bit(stop);
            if (pos == cap) {
                if (cap > mlen + cap_syms) return;          // runaway
                cap += cap / 2;
                c.buf.resize(cap + HIST);
                base = c.buf.data() + HIST;
            }
            const size_t got = z.run(base + pos, cap - pos);
            if (z.error()) return;
            size_t from = 0;                                // pieces: cut at the member ends reported for this call
            for (const auto &e : z.member_ends()) {
                crc = snk::crc32_fast(crc, base + pos + from, e.out_off - from);
                c.pieces.push_back(Piece{(uint64_t)(pos + e.out_off - piece_from), crc, true, e.crc, e.isize});
                crc = 0;
                piece_from = pos + e.out_off;
                from = e.out_off;
            }
            crc = snk::crc32_fast(crc, base + pos + from, got - from);
            pos += got;
            if (z.stopped()) {
                if (z.bitpos() == stop) break;
                ++j;
                stop = next_stop(j, z.bitpos(), gave_up);
                if (gave_up) return;
                continue;
            }
            if (z.done()) { stop = ~0ull; break; }
            if (got == 0 && pos < cap) return;              // no progress without a reason
        }
        if (pos > piece_from || c.pieces.empty()) c.pieces.push_back(Piece{(uint64_t)(pos - piece_from), crc, false, 0, 0});
        finish(c, z, pos - mlen, stop == ~0ull);
    }
    void finish(Chunk &c, MarkerInflate &z, size_t byte_len, bool to_end) {
        c.out_len = c.mlen + byte_len;
        c.end_bit = z.bitpos();
        c.whole_tail = to_end;
        c.ok = true;
        if (getenv("SNK_PGZ_DEBUG")) fprintf(stderr, "chunk %zu: start %llu end %llu markers %zu bytes %zu\n", (size_t)(&c - chunks_.get()),
                                             (unsigned long long)c.start_bit, (unsigned long long)c.end_bit, c.mlen, byte_len);
    }
    // all markers of a chunk -> bytes (the chain thread supplied the window), CRC-32 of that front part
    void resolve_chunk(Chunk &c) {
        if (c.buf.empty()) take(c.buf);
        if (c.buf.size() < c.out_len + HIST) c.buf.resize(c.out_len + HIST);
        uint8_t *base = c.buf.data() + HIST;
        const uint16_t *M = c.M.data();
        const uint8_t *w = c.win.data();
        size_t k = 0;
#if defined(__SSE2__)
        // sixteen symbols at a time: most groups hold no marker (in FASTQ the markers live in the read names) and are just
        // narrowed; a group with one takes the scalar look-ups
        for (; k + 16 <= c.mlen; k += 16) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(M + k));
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(M + k + 8));
            const __m128i hi = _mm_srli_epi16(_mm_or_si128(a, b), 8);
            if (_mm_movemask_epi8(_mm_cmpeq_epi16(hi, _mm_setzero_si128())) == 0xFFFF) {
                _mm_storeu_si128(reinterpret_cast<__m128i *>(base + k), _mm_packus_epi16(a, b));
            } else {
                for (size_t t = k; t < k + 16; ++t) {
                    const uint16_t v = M[t];
                    base[t] = v < 256 ? (uint8_t)v : w[v - 256];
                }
            }
        }
#endif
        for (; k < c.mlen; ++k) {
            const uint16_t v = M[k];
            base[k] = v < 256 ? (uint8_t)v : w[v - 256];
        }
        c.crc_front = snk::crc32_fast(0, base, c.mlen);
        give(c.M);
        std::vector<uint8_t>().swap(c.win);
    }

    // ---------------------------------------------------------------- chain thread
    void wait_done(size_t j) {
        std::unique_lock<std::mutex> l(m_);
        if (j > chain_need_) { chain_need_ = j; cv_.notify_all(); }      // the workers' window follows the chain
        cv_.wait(l, [&] { return quit_ || chunks_[j].done.load(std::memory_order_acquire) != 0; });
    }
    void drop(size_t j) {                                   // a chunk that is not part of the stream (no start / false start)
        wait_done(j);
        give(chunks_[j].M);
        give(chunks_[j].buf);
    }
    void chain() {
        std::vector<uint8_t> win(HIST, 0), nxt(HIST);
        size_t cur = first_;
        uint64_t expect = mid_ ? start_bit_ : 0;
        if (mid_) win = win0_;
        for (;;) {
            wait_done(cur);
            { std::lock_guard<std::mutex> l(m_); if (quit_) return; }
            Chunk &c = chunks_[cur];
            if (!c.ok) { break_chain(expect, cur == first_ && !mid_, win); return; }
            // the window behind this chunk: its last 32 KiB (with the tail of the old window when it is shorter)
            {
                const size_t n = c.out_len, take = std::min<size_t>(n, HIST);
                if (take < HIST) memcpy(nxt.data(), win.data() + take, HIST - take);
                const uint8_t *bytes = c.buf.size() >= HIST ? c.buf.data() + HIST : nullptr;
                for (size_t t = 0; t < take; ++t) {
                    const size_t k = n - take + t;
                    uint8_t b;
                    if (k < c.mlen) { const uint16_t v = c.M[k]; b = v < 256 ? (uint8_t)v : win[v - 256]; }
                    else b = bytes[k];
                    nxt[HIST - take + t] = b;
                }
            }
            c.win = win;
            win.swap(nxt);
            {
                std::lock_guard<std::mutex> l(m_);
                resolve_q_.push_back(cur);
                order_.push_back(cur);
            }
            cv_.notify_all();
            if (c.whole_tail) {
                { std::lock_guard<std::mutex> l(m_); chain_end_ = true; }
                cv_.notify_all();
                // chunks behind the end (trailing garbage): let the workers finish, nothing to keep
                return;
            }
            expect = c.end_bit;
            size_t j = cur + 1;
            for (; j < nchunks_; ++j) {
                if (ensure_search(j) == expect) break;
                drop(j);
                { std::lock_guard<std::mutex> l(m_); if (quit_) return; skipped_.push_back(j); }
            }
            if (j == nchunks_) { break_chain(expect, false, win); return; }
            cur = j;
        }
    }
    void break_chain(uint64_t bit, bool from_header, const std::vector<uint8_t> &win) {
        std::lock_guard<std::mutex> l(m_);
        fb_bit_ = bit;
        fb_header_ = from_header;
        fb_win_ = win;
        chain_fallback_ = true;
        cv_.notify_all();
    }
    std::vector<size_t> skipped_;
    // chunk buffers are recycled (a fresh 10-20 MB vector per chunk is 10-20 MB of page faults per chunk)
    std::mutex bm_;
    std::vector<std::vector<uint16_t>> free16_;
    std::vector<std::vector<uint8_t>> free8_;
    void take(std::vector<uint16_t> &v) { std::lock_guard<std::mutex> l(bm_); if (!free16_.empty()) { v.swap(free16_.back()); free16_.pop_back(); } }
    void take(std::vector<uint8_t> &v) { std::lock_guard<std::mutex> l(bm_); if (!free8_.empty()) { v.swap(free8_.back()); free8_.pop_back(); } }
    void give(std::vector<uint16_t> &v) {
        if (!v.capacity()) return;
        std::lock_guard<std::mutex> l(bm_);
        if (free16_.size() < 2 * lookahead_) { free16_.emplace_back(); free16_.back().swap(v); } else std::vector<uint16_t>().swap(v);
    }
    void give(std::vector<uint8_t> &v) {
        if (!v.capacity()) return;
        std::lock_guard<std::mutex> l(bm_);
        if (free8_.size() < 2 * lookahead_) { free8_.emplace_back(); free8_.back().swap(v); } else std::vector<uint8_t>().swap(v);
    }

    // ---------------------------------------------------------------- consumer
    void fail(const char *msg) { err_ = msg; }
    void add_piece(uint32_t crc, uint64_t len) {
        mcrc_ = (uint32_t)crc32_combine(mcrc_, crc, (z_off_t)len);
        mlen_ += len;
    }
    bool end_member(uint32_t want_crc, uint32_t want_isize) {
        if (partial_) { partial_ = false; mcrc_ = 0; mlen_ = 0; return true; }      // the member the start lay in: its front was not seen
        if (mcrc_ != want_crc) { fail("gzip CRC mismatch"); return false; }
        if ((uint32_t)mlen_ != want_isize) { fail("gzip length mismatch"); return false; }
        mcrc_ = 0;
        mlen_ = 0;
        return true;
    }
    void release_current() {
        Chunk &c = chunks_[cur_];
        give(c.buf);
        cur_ready_ = false;
        {
            std::lock_guard<std::mutex> l(m_);
            // everything up to the next chunk of the chain is done with
            size_t upto = cur_ + 1;
            if (order_pos_ < order_.size()) upto = order_[order_pos_];
            if (upto > consumed_) consumed_ = upto;
        }
        cv_.notify_all();
    }
    // makes the next chunk of the chain ready for copying out; false: switched to the fallback / finished / failed
    bool next_chunk() {
        size_t i;
        {
            std::unique_lock<std::mutex> l(m_);
            cv_.wait(l, [&] { return order_pos_ < order_.size() || chain_end_ || chain_fallback_; });
            if (order_pos_ == order_.size()) {
                if (chain_fallback_) {
                    l.unlock();
                    win_ = fb_win_;
                    seq_from(fb_bit_, fb_header_);
                } else {
                    done_ = true;
                }
                return false;
            }
            i = order_[order_pos_++];
            if (i + 1 > consumed_) consumed_ = i;           // (skipped chunks in front of i are done with)
            cv_.wait(l, [&] { return chunks_[i].resolved.load(std::memory_order_acquire) != 0; });
        }
        cv_.notify_all();
        cur_ = i;
        Chunk &c = chunks_[i];
        if (on_chunk) on_chunk(c.start_bit, handed_out_);
        add_piece(c.crc_front, c.mlen);
        for (const Piece &p : c.pieces) {
            add_piece(p.crc, p.len);
            if (p.ends_member && !end_member(p.want_crc, p.want_isize)) return false;
        }
        cur_off_ = 0;
        cur_ready_ = true;
        if (c.out_len == 0) release_current();
        return cur_ready_;
    }

    // ---------------------------------------------------------------- sequential fallback
    void seq_from(uint64_t bit, bool from_header) {
        if (getenv("SNK_PGZ_DEBUG")) fprintf(stderr, "sequential from bit %llu\n", (unsigned long long)bit);
        seq_ = true;
        sq_.init(in_, n_);
        sq_.set_verify_crc(false);
        if (!from_header) sq_.start_at_block(bit);
        sbuf_.assign(HIST + ((size_t)1 << 22), 0);
        memcpy(sbuf_.data(), win_.data(), HIST);
        s_have_ = s_off_ = 0;
    }
    size_t seq_run(uint8_t *out, size_t cap) {
        if (s_off_ == s_have_) {
            if (s_have_) memmove(sbuf_.data(), sbuf_.data() + s_have_, HIST);      // keep the window in front
            uint8_t *p = sbuf_.data() + HIST;
            s_have_ = sq_.run(p, sbuf_.size() - HIST);
            s_off_ = 0;
            if (sq_.error()) { fail(sq_.error()); return 0; }
            size_t from = 0;
            for (const auto &e : sq_.member_ends()) {
                add_piece(snk::crc32_fast(0, p + from, e.out_off - from), e.out_off - from);
                if (!end_member(e.crc, e.isize)) return 0;
                from = e.out_off;
            }
            add_piece(snk::crc32_fast(0, p + from, s_have_ - from), s_have_ - from);
            if (s_have_ == 0 && sq_.done()) { done_ = true; return 0; }
        }
        const size_t n = std::min(cap, s_have_ - s_off_);
        memcpy(out, sbuf_.data() + HIST + s_off_, n);
        s_off_ += n;
        return n;
    }
};

}  // namespace snk
#endif
